cd $GRAFT_REPO_ROOT
WAE_GMRES_DEBUG=1 timeout -k 10 500 python dev/newton_c3.py C3 8 > gpurun_out/r3_newton.log 2>&1
grep -v "^\[gmres\] nb=\(16\|32\|64\)" gpurun_out/r3_newton.log | grep -v "^\[rb\]" | tail -n 60
