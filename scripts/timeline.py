"""Gaps between consecutive kernels of the LM chain from a rocprofv3 --kernel-trace CSV."""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void movba::", "").replace("movba::", ""), r.get("Queue_Id", "")) for r in rows]
# last solve only: everything after the last k_init_pose
last = max(i for i, k in enumerate(ks) if "k_init_pose" in k[2])
ks = ks[last:]
t0 = ks[0][0]
busy = defaultdict(float); gap_after = defaultdict(list)
main = [k for k in ks if "k_coarse" not in k[2]]
for a, b in zip(main, main[1:]):
    gap_after[a[2] + " -> " + b[2]].append((b[0] - a[1]) / 1e3)
for k in ks: busy[k[2]] += (k[1] - k[0]) / 1e3
print("span %.1f us, kernels (main stream) busy %.1f us" % ((main[-1][1] - t0) / 1e3, sum((k[1] - k[0]) for k in main) / 1e3))
for k, v in busy.items(): print("  busy %-28s %8.1f us" % (k, v))
for k, v in gap_after.items(): print("  gap  %-44s n=%3d mean %6.2f us  sum %7.1f" % (k, len(v), sum(v) / len(v), sum(v)))
if len(sys.argv) > 2:
    for k in ks[:60]: print("%9.2f %9.2f %8.2f %s q%s" % ((k[0] - t0) / 1e3, (k[1] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[2], k[3]))
