"""GPU-side timeline of ONE whole movba_lba_solve call from a rocprofv3 --kernel-trace --memory-copy-trace run of
scripts/upload_laps.py (the last call): every copy and kernel with start / end relative to the call's first GPU activity.
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 scripts/upload_laps.py cfg3
    python3 scripts/call_timeline.py DIR [n_rows]"""
import csv, glob, sys
d = sys.argv[1]
nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 60
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(kt)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void movba::", "").replace("movba::", "")[:60]))
if mc:
    for r in csv.DictReader(open(mc[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
last = max(i for i, e in enumerate(ev) if "k_init_pose" in e[2])
# the call's first activity: walk back from k_init_pose to the end of the previous call (its k_export)
i0 = last
while i0 > 0 and "k_export" not in ev[i0 - 1][2]: i0 -= 1
t0 = ev[i0][0]
end = max(e[1] for e in ev[i0:])
print("call span on the GPU: %.1f us" % ((end - t0) / 1e3))
for e in ev[i0:i0 + nrows]:
    print("%9.2f %9.2f %8.2f  %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2]))
print("...")
for e in ev[-8:]:
    print("%9.2f %9.2f %8.2f  %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2]))
