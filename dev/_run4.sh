cd $GRAFT_REPO_ROOT
WAE_SETUP_DEBUG=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-newton > gpurun_out/r3_setup2.json 2> gpurun_out/r3_setup2.err
grep "^\[setup\]\|^\[amg\]\|^\[create\]" gpurun_out/r3_setup2.err
