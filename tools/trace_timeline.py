import csv, re, sys
f=sys.argv[1]; want=int(sys.argv[2])
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
step=-1; t0=None; prev_end={}
for r in rows:
    n=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")
    if n.startswith("k_colsum4"): step+=1
    if step==want-1 and t0 is None and n.startswith("k_match_bf3"): t0=int(r["Start_Timestamp"])
    if t0 is None: continue
    if step>want: break
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print("%9.1f %9.1f %8.1f  q%s s%s  %s"%(s,e,e-s,r["Queue_Id"],r["Stream_Id"],n[:50]))
