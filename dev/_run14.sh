cd $GRAFT_REPO_ROOT
for CFG in "0 0" "0 1" "0 2" "0 6" "1 6" "2 1" "2 2" "3 2" "2 0"; do
  set -- $CFG
  WAE_BENCH_TC=$1 WAE_BENCH_MODE=$2 timeout -k 10 200 python dev/spmv_only.py C3 64 2>&1 | tail -n 1 | sed "s/^/tc $1 mode $2: /"
done
