import csv, collections, os
rows=list(csv.DictReader(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+'/gpurun_out/pipe_trace/run_kernel_trace.csv')))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
sp=[i for i,r in enumerate(rows) if "k_points_fast" in r["Kernel_Name"]]
seg=rows[sp[-101]:sp[-1]]
t0,t1=int(seg[0]["Start_Timestamp"]),int(seg[-1]["End_Timestamp"])
q=collections.defaultdict(lambda:[0,0.0,set()])
for r in seg:
    k=r["Queue_Id"]
    q[k][0]+=1; q[k][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    q[k][2].add(r["Kernel_Name"].split("(")[0].replace("void ","").replace("ksk::","").replace("ksrs::","")[:14])
print("window us/frame", (t1-t0)/1e5)
for k,v in sorted(q.items()): print("queue",k,"dispatches",v[0],"busy %.1f%%"%(100*v[1]*1e3/(t1-t0)), "us/frame %.0f"%(v[1]/100), sorted(v[2]))
