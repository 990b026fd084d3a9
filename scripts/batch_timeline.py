"""Start/end of every kernel of the last batched run in a rocprofv3 --kernel-trace CSV (two streams side by side)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void movba::", "").replace("movba::", ""), r.get("Stream_Id", r.get("Queue_Id", ""))) for r in rows]
inits = [i for i, k in enumerate(ks) if "k_init_pose" in k[2]]
ks = ks[inits[-2] if len(inits) >= 2 else inits[-1]:]
t0 = ks[0][0]
for k in ks[:int(sys.argv[2]) if len(sys.argv) > 2 else 70]:
    print("%9.2f %9.2f %8.2f  s%-3s %s" % ((k[0] - t0) / 1e3, (k[1] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], k[2]))
print("span %.1f us" % ((ks[-1][1] - t0) / 1e3))
