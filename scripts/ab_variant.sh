for i in 1 2; do
echo prev; MOVBA_LIB=$GRAFT_REPO_ROOT/build/libmovba_prev.so timeout -k 10 100 python scripts/variant_time.py cfg3 2>&1 | tail -3
echo new; timeout -k 10 100 python scripts/variant_time.py cfg3 2>&1 | tail -3
done
