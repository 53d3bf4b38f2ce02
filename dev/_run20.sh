cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_tile_parity.py tests/test_gpu_parity.py -x -q -m gpu -k "tile or long_rows or solve or householder or newton or G1 or G5 or perturb" > gpurun_out/r3_t5.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t5.log
tail -n 6 gpurun_out/r3_t5.log
WAE_GMRES_DEBUG=0 timeout -k 10 400 python dev/newton_c3.py C3 8 2>&1 | tail -n 2
