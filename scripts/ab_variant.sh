#!/bin/bash
# A/B on ONE box: scripts/variant_time.py with build/libmovba_prev.so (a build of an earlier commit: git worktree add /tmp/wt <commit>;
# make -C /tmp/wt/mov-slam_amd/csrc; cp /tmp/wt/mov-slam_amd/libmovba.so build/libmovba_prev.so) and with the current library, twice each.
# Box-to-box spread (3 - 5 %) is larger than most single changes: compare on the same box.
for i in 1 2; do
echo prev; MOVBA_LIB=$GRAFT_REPO_ROOT/build/libmovba_prev.so timeout -k 10 100 python scripts/variant_time.py cfg3 2>&1 | tail -3
echo new; timeout -k 10 100 python scripts/variant_time.py cfg3 2>&1 | tail -3
done
