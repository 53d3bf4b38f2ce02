cd $GRAFT_REPO_ROOT
for E in 2 0.5 0.25 0.1; do
WAE_REORTH_ETA=$E WAE_GMRES_DEBUG=1 timeout -k 10 400 python dev/newton_c3.py C3 8 > gpurun_out/r3_newton_eta.log 2>&1
echo "eta=$E $(grep householder_many gpurun_out/r3_newton_eta.log) its: $(grep 'nb=[82] x0=0' gpurun_out/r3_newton_eta.log | sed 's/.*lockstep_its=\([0-9]*\).*/\1/' | tr '\n' ' ')"
grep "^[0-9] [0-9]" gpurun_out/r3_newton_eta.log | awk '{print $NF}' | tr '\n' ' '; echo
done
