#!/usr/bin/env python3
"""Turn the rocprofv3 output of scripts/profile_gpu.sh into the small summaries kept under profiles/.

    python scripts/summarise_profiles.py gpurun_out/prof_<tag> <tag>

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3 --stats) and profiles/<tag>_pmc_traffic.json:
per kernel, launches with real work only (no-op run-ahead launches fetch ~nothing and would dilute the
average), mean FETCH_SIZE / WRITE_SIZE per launch in bytes.  FETCH_SIZE is reported in KiB of 64-B
requests; MI355X_MICROARCH.md (HBM section) says it counts exactly half of the bytes of a wide coalesced
stream on gfx950 — the raw and the doubled figure are kept, and the CORRECTED one: fetch / (1 - s/2) with s the share of the
kernel's load bytes moved by 16-byte-per-lane loads (ISA count, scripts/isa_load_widths.py); bench.py quotes the corrected one.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
st = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
if st:
    shutil.copy(st[0], os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))


def per_kernel(path, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(src, path, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
# share of each kernel's load bytes that 16-byte-per-lane loads account for (scripts/isa_load_widths.py): FETCH_SIZE counts
# those at half their bytes (MI355X_MICROARCH.md, HBM section), narrower loads at face value
widths = {}
wf = sorted(glob.glob(os.path.join(root, "profiles", "*_isa_load_widths.json")))
if wf:
    widths = {k: v["share_of_load_bytes_16B"] for k, v in json.load(open(wf[-1]))["kernels"].items()}
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    thr = 0.05 * max(f) if f else 0.0
    fw = [v for v in f if v > thr] or [0.0]
    thrw = 0.05 * max(w) if w else 0.0
    ww = [v for v in w if v > thrw] or [0.0]
    out[k] = {"launches_total": len(f), "launches_with_work": len(fw),
              "fetch_kib_mean": sum(fw) / len(fw), "write_kib_mean": sum(ww) / len(ww),
              "hbm_bytes_per_launch_raw": 1024 * (sum(fw) / len(fw) + sum(ww) / len(ww)),
              "hbm_bytes_per_launch_fetch_doubled": 1024 * (2 * sum(fw) / len(fw) + sum(ww) / len(ww))}
    share = next((v for n, v in widths.items() if n == k or n.split("(")[0] == k.split("(")[0]), None)
    if share is not None:
        out[k]["share_of_load_bytes_16B"] = share
        out[k]["hbm_bytes_per_launch_corrected"] = 1024 * ((sum(fw) / len(fw)) / (1.0 - 0.5 * share) + sum(ww) / len(ww))
json.dump({"tag": tag, "units": "FETCH_SIZE/WRITE_SIZE are KiB per dispatch (rocprofv3 --pmc, separate passes)", "kernels": out},
          open(os.path.join(root, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print(f"{k[:48]:48s} work-launches {v['launches_with_work']:4d}  fetch {v['fetch_kib_mean']:9.1f} KiB  write {v['write_kib_mean']:9.1f} KiB")
