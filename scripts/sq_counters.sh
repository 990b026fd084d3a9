#!/bin/bash
# sq_counters.sh <tag> <name> <script> [args...]: SQ wave-cycle counters of one Python target (rocprofv3 --pmc, its own run:
# no trace options beside it), summarised per kernel into profiles/<tag>_sq_counters_<name>.json by scripts/summarise_sq.py.
# Run through gpurun from the repo root.
TAG=$1; NAME=$2; shift 2
# the target: a script under scripts/, or (with a slash) a path from the repo root, e.g. tests/dev/pattern_time.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_${TAG}_$NAME; mkdir -p $OUT
case $1 in */*) REL=$1;; *) REL=scripts/$1;; esac; TARGET=$ROOT/$REL; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
    --output-format csv -d $OUT/pmc -- python3 $TARGET "$@" > $OUT/run.log 2>&1 || exit 1
cd $ROOT && python3 scripts/summarise_sq.py $OUT/pmc $TAG $NAME "$REL $*"
# the summary is written on the GPU box: only gpurun_out/ comes back, so a copy goes there (commit it under profiles/ afterwards)
cp $ROOT/profiles/${TAG}_sq_counters_$NAME.json $OUT/
