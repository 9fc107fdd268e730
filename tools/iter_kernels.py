import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("render_bwd_kernel")]
# pick an iteration in the middle of the timed region
a,b=idx[len(idx)//2], idx[len(idx)//2+1]
t_prev=int(rows[a]["End_Timestamp"])
tot=0; gaps=0
for r in rows[a+1:b+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-58s dur %7.1f us  gap before %6.1f us" % (r["Kernel_Name"].split("(")[0][:58],(e-s)/1e3,(s-t_prev)/1e3))
    tot+=(e-s); gaps+=max(0,s-t_prev); t_prev=e
print("kernels %.1f us, gaps %.1f us, n=%d" % (tot/1e3,gaps/1e3,b-a))
