#!/bin/bash
# ab_libs.sh <cfg> <lib|-> ... : scripts/variant_time.py with each library in turn (- = the tree's own), two rounds, on ONE box
# (box-to-box spread is larger than most single changes).  A library named `x@VAR=1` runs with that environment variable set.
CFG=$1; shift
for i in 1 2; do
  for L in "$@"; do
    ENVV=""; LIB=$L
    case $L in *@*) ENVV=${L#*@}; LIB=${L%@*};; esac
    if [ "$LIB" = "-" ]; then LIBP=""; else LIBP=$GRAFT_REPO_ROOT/build/libmovba_$LIB.so; fi
    echo -n "$L: "; env $ENVV MOVBA_LIB=$LIBP timeout -k 10 100 python scripts/variant_time.py $CFG 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
