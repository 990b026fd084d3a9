#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_* output -> profiles/<tag>_sq_counters_<name>.json: per kernel, the mean over the upper half of its
launches by SQ_WAVE_CYCLES (launches with work: the LM loop's run-ahead launches are no-ops), fractions of SQ_WAVE_CYCLES.
WAIT_ANY = waves parked at s_waitcnt / barriers, WAIT_INST_ANY = issue stalls (dependencies, pipes), ACTIVE_INST_ANY = issuing
(MI355X_MICROARCH.md, PMC section)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

src, tag, name, cmd = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = defaultdict(lambda: defaultdict(dict))          # kernel -> dispatch id -> counter -> value
for f in glob.glob(os.path.join(src, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
out = {}
for k, disp in rows.items():
    ds = sorted(disp.values(), key=lambda d: d.get("SQ_WAVE_CYCLES", 0.0))
    ds = ds[len(ds) // 2:]
    mean = lambda c: sum(d.get(c, 0.0) for d in ds) / len(ds)
    wc = mean("SQ_WAVE_CYCLES")
    short = re.sub(r"^void ", "", k).split("(")[0].replace("movba::", "").replace("(anonymous namespace)::", "")
    e = {"launches": len(disp), "SQ_WAVE_CYCLES": round(wc), "SQ_INSTS_VALU": round(mean("SQ_INSTS_VALU"))}
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"):
        e[c + "_frac"] = round(mean(c) / wc, 4) if wc else 0.0
    out[short] = e
path = os.path.join(root, "profiles", f"{tag}_sq_counters_{name}.json")
json.dump({"tag": tag, "command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU "
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -- python3 scripts/" + cmd,
           "note": __doc__.split("\n", 1)[1].strip(), "kernels": out}, open(path, "w"), indent=1)
for k, e in sorted(out.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:8]:
    print(f"{k[:48]:48s} issuing {e['SQ_ACTIVE_INST_ANY_frac']:.2f} (valu {e['SQ_ACTIVE_INST_VALU_frac']:.2f}) stalled {e['SQ_WAIT_INST_ANY_frac']:.2f} parked {e['SQ_WAIT_ANY_frac']:.2f}")
print("wrote", path)
