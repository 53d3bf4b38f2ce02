#!/bin/bash
# The round's profiles/ entries.  Two GPU calls:  dev/final_round.sh spmv TAG   then (after dev/make_traffic_json.py TAG has
# written profiles/<TAG>_spmv_traffic_*.json here)   dev/final_round.sh bench TAG
set -e
WHAT=${1:-spmv}; TAG=${2:-r03}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$WHAT" = spmv ]; then
    timeout -k 10 420 dev/collect_spmv_profile.sh C3 ${TAG}_C3 > $OUT/spmv_C3.log 2>&1
    timeout -k 10 170 dev/collect_spmv_profile.sh C2 ${TAG}_C2 > $OUT/spmv_C2.log 2>&1
else
    # (the C3 bench runs FIRST: its value_cold is that of the first process on a freshly restored box, without any prefault -- round 4)
    timeout -k 10 500 python bench.py > $OUT/bench.log 2> $OUT/bench.err
    grep "^{" $OUT/bench.log > $OUT/bench.json
    timeout -k 10 300 python bench.py --preset C2 --N 32 --l 16 > $OUT/bench_C2.log 2> $OUT/bench_C2.err
    grep "^{" $OUT/bench_C2.log > $OUT/bench_C2.json
    timeout -k 10 300 dev/prof_bench.sh $OUT/prof_bench > $OUT/prof_bench.log 2>&1
fi
echo done
