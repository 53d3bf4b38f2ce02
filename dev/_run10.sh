cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_t3.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t3.log
tail -n 6 gpurun_out/r3_t3.log
