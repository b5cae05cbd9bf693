import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+"/*/*kernel_trace.csv"))[-1]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if "k_wf" in r["Kernel_Name"] or "k_render" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
n=int(sys.argv[2]) if len(sys.argv)>2 else 32
prev_end=None
for r in rows[:n]:
    nm=r["Kernel_Name"].split("(")[0].replace("void jtx::","")
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    gap = (s-prev_end)/1e3 if prev_end else 0
    prev_end=e
    print("%10.1fus gap %6.1f dur %9.1fus  %-34s vgpr %s sgpr %s scr %s" % ((s-t0)/1e3,gap,(e-s)/1e3,nm[:34],r.get("VGPR_Count"),r.get("SGPR_Count"),r.get("Scratch_Size")))
