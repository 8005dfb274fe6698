import csv,glob,sys
d=sys.argv[1]
f=glob.glob(d+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'render_bwd' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['End_Timestamp']); 
prev_end=None; busy=0; gaps=0
for i in range(a+1,b+1):
    r=rows[i]; nm=r['Kernel_Name'].split('(')[0][-48:]
    d_=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    gap=(int(r['Start_Timestamp'])-prev_end)/1e3 if prev_end else (int(r['Start_Timestamp'])-t0)/1e3
    prev_end=int(r['End_Timestamp']); busy+=d_; gaps+=max(gap,0)
    if len(sys.argv)>2: print(f"{nm:48s} {d_:8.1f} us  gap {gap:6.1f}")
print("step span us", (int(rows[b]['End_Timestamp'])-t0)/1e3, "busy", busy, "gaps", gaps, "launches", b-a)
