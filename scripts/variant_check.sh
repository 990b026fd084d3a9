#!/bin/bash
# variant_check.sh <lib> <log>: one forced-direct cfg3 solve on a library variant; fails on a runtime fault or wrong exit
MOVBA_LIB=$1 timeout -k 10 100 python scripts/dense_stamps.py cfg3 > $2 2>&1
rc=$?
if grep -q "Memory access fault\|core dump\|Aborted" $2; then echo "FAULT in $1"; exit 1; fi
tail -1 $2
exit $rc
