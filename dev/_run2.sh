cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/r3_t2.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t2.log
for P in -1 2 4 6 8; do
  WAE_GMRES_PAIR=$P timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-newton > gpurun_out/r3_b1_pair$P.json 2> gpurun_out/r3_b1_pair$P.err; echo "pair $P rc=$?" >> gpurun_out/r3_t2.log
done
tail -n 5 gpurun_out/r3_t2.log
