#!/bin/bash
# rocprofv3 kernel trace + stats of one window solve loop per covisibility pattern, of cfg3 on the one-launch direct solver, of
# the larger windows (tests/dev/direct_time.py), of the pose kernels and of a batched run of eight windows
# (run through gpurun from the repo root):  gpurun_out/prof_<tag>_<name>/stats -> profiles/<tag>_<name>_kernel_stats.csv
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for name in cfg3 shuffled revisit hub; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_$name; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tests/dev/pattern_time.py $name > $OUT/stats.log 2>&1 || exit 1
  cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $ROOT/gpurun_out/${TAG}_${name}_kernel_stats.csv || exit 1
done
OUT=$ROOT/gpurun_out/prof_${TAG}_cfg3_direct; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tests/dev/pattern_time.py cfg3 --direct > $OUT/stats.log 2>&1 || exit 1
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $ROOT/gpurun_out/${TAG}_cfg3_direct_kernel_stats.csv || exit 1
OUT=$ROOT/gpurun_out/prof_${TAG}_big_direct; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tests/dev/direct_time.py > $OUT/stats.log 2>&1 || exit 1
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $ROOT/gpurun_out/${TAG}_direct_time_kernel_stats.csv || exit 1
tail -6 $OUT/stats.log
# the pose kernels (movba_pose_opt, cfg1) and the batched run of eight resident windows
OUT=$ROOT/gpurun_out/prof_${TAG}_pose; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tests/dev/pose_time.py > $OUT/stats.log 2>&1 || exit 1
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $ROOT/gpurun_out/${TAG}_pose_kernel_stats.csv || exit 1
OUT=$ROOT/gpurun_out/prof_${TAG}_batch8; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/scripts/batch_time.py 8 > $OUT/stats.log 2>&1 || exit 1
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $ROOT/gpurun_out/${TAG}_batch8_kernel_stats.csv || exit 1
tail -2 $OUT/stats.log
# HBM traffic of the batched kernels (one --pmc pass per counter, nothing else beside it):
# scripts/summarise_profiles.py gpurun_out/prof_<tag>_batch8 <tag>_batch8 -> profiles/<tag>_batch8_pmc_traffic.json
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/scripts/batch_time.py 8 --short > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/scripts/batch_time.py 8 --short > $OUT/pmc_write.log 2>&1 || exit 1
