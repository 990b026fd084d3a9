#!/bin/bash
# rocprofv3 kernel stats of the structure-pass kernels with one phase compiled out (run through gpurun from the repo root after
#   bash scripts/build_variant.sh skip1 -DMOVBA_STRUCT_SKIP=1   (no pair-mask loop)
#   bash scripts/build_variant.sh skip2 -DMOVBA_STRUCT_SKIP=2   (no entry-fill loop)
# ): what the setup, the mask phase and the fill phase of k_struct_pairs each cost.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base skip1 skip2; do
  if [ $v = base ]; then unset MOVBA_LIB; else export MOVBA_LIB=$R/build/libmovba_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_skip_$v -- python3 $R/scripts/upload_only.py > $R/gpurun_out/prof_skip_$v.log 2>&1 || exit 1
  echo "== $v"; grep "struct" $R/gpurun_out/prof_skip_$v/*/*kernel_stats.csv | cut -c1-110
done
