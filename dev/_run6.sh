cd $GRAFT_REPO_ROOT
WAE_GMRES_DEBUG=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-newton > gpurun_out/r3_first.json 2> gpurun_out/r3_first.err
grep -n "^\[rb\]" gpurun_out/r3_first.err
grep "^\[gmres\]" gpurun_out/r3_first.err | awk '{print NR": "$0}' | head -80
