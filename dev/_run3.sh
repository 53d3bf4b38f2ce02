cd $GRAFT_REPO_ROOT
WAE_SETUP_DEBUG=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-newton > gpurun_out/r3_setup.json 2> gpurun_out/r3_setup.err
timeout -k 10 400 dev/prof_bench.sh gpurun_out/r3_prof1 > gpurun_out/r3_prof1.log 2>&1
tail -n 25 gpurun_out/r3_prof1.log
