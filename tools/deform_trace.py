import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
fw = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'deform_fwd_kernel' in r['Kernel_Name']]
bw = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'deform_bwd_kernel' in r['Kernel_Name']]
med = lambda v: sorted(v)[len(v) // 2] if v else float('nan')
for i in range(len(fw) // 110):
    print("variant %d: deform_fwd median %.1f us, deform_bwd median %.1f us" % (i, med(fw[i * 110:(i + 1) * 110]), med(bw[i * 55:(i + 1) * 55])))
