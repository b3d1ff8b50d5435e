#!/usr/bin/env python3
"""Where does the GPU idle inside the timed region?  Largest inter-kernel gaps of a kernel trace."""
import csv, collections, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
ev=sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows])
names=[e[2] for e in ev]
steps=int(sys.argv[2]) if len(sys.argv)>2 else 40
damp=[i for i,n in enumerate(names) if n.startswith('k_lm_damp') or n.startswith('k_lm_lsmr_setup')]
# the timed region = the 40 steps after the 8 warm-up steps
start=damp[8]; end=damp[8+steps] if len(damp)>8+steps else len(ev)-1
seg=ev[start:end]
busy=sum(e[1]-e[0] for e in seg); span=seg[-1][1]-seg[0][0]
print("kernels %d  busy %.1f us/step  span %.1f us/step"%(len(seg), busy/1e3/steps, span/1e3/steps))
gaps=collections.defaultdict(lambda:[0,0.0])
for a,b in zip(seg,seg[1:]):
    k=(a[2].split('(')[0][:44], b[2].split('(')[0][:44]); gaps[k][0]+=1; gaps[k][1]+=(b[0]-a[1])/1e3
for k,v in sorted(gaps.items(), key=lambda kv:-kv[1][1])[:16]:
    print("%8.1f us/step %5.1f x/step avg %6.2f   %s -> %s"%(v[1]/steps,v[0]/steps,v[1]/v[0],k[0],k[1]))
