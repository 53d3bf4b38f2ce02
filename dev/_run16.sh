cd $GRAFT_REPO_ROOT
for M in 150 80 50 30; do
WAE_GMRES_NARROW_M=$M WAE_GMRES_DEBUG=1 timeout -k 10 400 python dev/newton_c3.py C3 8 > gpurun_out/r3_newton_$M.log 2>&1
echo "M=$M $(grep householder_many gpurun_out/r3_newton_$M.log) its: $(grep 'nb=8 x0=0' gpurun_out/r3_newton_$M.log | sed 's/.*lockstep_its=\([0-9]*\).*/\1/' | tr '\n' ' ')"
done
