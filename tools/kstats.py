import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
steps=float(sys.argv[2]) if len(sys.argv)>2 else 23
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 22]:
    n=int(r['Calls']); t=float(r['TotalDurationNs'])
    print(f"{r['Name'][:78]:78s} {n:5d} {t/steps/1e6:8.4f} ms/step avg {float(r['AverageNs'])/1e3:8.1f} us")
print("total ms/step", sum(float(r['TotalDurationNs']) for r in rows)/steps/1e6)
