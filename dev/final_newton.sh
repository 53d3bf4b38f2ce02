#!/bin/bash
# The round-end artefacts in one GPU call: dev/final_round.sh bench TAG, then the kernel stats of the Newton-type phase (dev/newton_c3.py).
bash dev/final_round.sh bench r03b || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final_r03b/prof_newton -o nw -- python3 $GRAFT_REPO_ROOT/dev/newton_c3.py C3 8 > $GRAFT_REPO_ROOT/gpurun_out/final_r03b/prof_newton.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/final_r03b -name "*kernel_trace.csv" -delete
find gpurun_out/final_r03b -name "*.db" -delete
echo final done
