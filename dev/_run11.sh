cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_parity.py -x -q -m gpu -k "distributed or mgpu or two_process or shape or unit_cell" > gpurun_out/r3_t4.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t4.log
tail -n 30 gpurun_out/r3_t4.log
