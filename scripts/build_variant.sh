#!/bin/bash
# build_variant.sh <tag> <extra -D flags...>: a diagnostic build of libmovba into build/libmovba_<tag>.so
set -e
TAG=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
B=/tmp/movba_variant_$TAG; mkdir -p $B $ROOT/build
cd $ROOT/mov-slam_amd/csrc
for f in kernels.hip pcg_kernel.hip band_kernel.hip dense_solve.hip dense_persist.hip struct_kernels.hip struct_sort.hip pose_kernels.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I$ROOT/include -I. -Wno-unused-function --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=on "$@" -c $f -o $B/${f%.hip}.o &
done
for f in api.cpp structure.cpp dense_plan.cpp pcg_plan.cpp; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I$ROOT/include -I. -Wno-unused-function --offload-arch=gfx950 -ffp-contract=on "$@" -x hip -c $f -o $B/${f%.cpp}.o &
done
FAIL=0; for j in $(jobs -p); do wait $j || FAIL=1; done
[ $FAIL = 0 ] || { echo "build_variant: a compile failed" >&2; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/libmovba_$TAG.so $B/*.o
echo built $ROOT/build/libmovba_$TAG.so
