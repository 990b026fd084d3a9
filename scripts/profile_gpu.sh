#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the default bench command      -> gpurun_out/prof_<tag>/stats
#   2. HBM traffic counters, one --pmc pass each (TCC slots: FETCH_SIZE and WRITE_SIZE do not fit together)
# Summaries are copied into profiles/ by scripts/summarise_profiles.py afterwards.
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1 || exit 1
ls -R $OUT | head -30
