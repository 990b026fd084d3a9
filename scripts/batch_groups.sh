for g in 2 3 4; do
  MOVBA_BATCH_GROUPS=$g timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/bench_g$g.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_g$g.json"))
print("groups $g", d["value"], d["config"]["batched_windows"]["ms_per_batch"], d["config"]["batched_windows"]["vs_one_resident_window_at_a_time"], d["config"]["resident_window"]["ms_per_window_solve"])
PY
done
