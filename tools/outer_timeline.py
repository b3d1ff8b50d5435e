import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_chol_chain' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['Start_Timestamp'])
prev_end=None
for r in rows[a:b+1]:
    st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
    gap=(st-prev_end)/1e3 if prev_end else 0
    print("%8.1f %7.1f gap %6.1f  %s" % ((st-t0)/1e3,(en-st)/1e3,gap,r['Kernel_Name'][:64]))
    prev_end=en
