cd $GRAFT_REPO_ROOT
WAE_VC_PRE=0 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-newton > gpurun_out/r3_vc.json 2> gpurun_out/r3_vc.err; echo rc=$?
tail -n 5 gpurun_out/r3_vc.err; head -c 600 gpurun_out/r3_vc.json
