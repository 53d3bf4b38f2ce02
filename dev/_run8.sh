cd $GRAFT_REPO_ROOT
WAE_TILE_WAVES=16 WAE_TILE_GRID=8 WAE_TILE_TAIL=4 timeout -k 10 600 python tests/tile_worker.py > gpurun_out/r3_quad_worker.log 2>&1; echo "worker rc=$?"
tail -n 3 gpurun_out/r3_quad_worker.log
for W in 8 16; do for R in 64 8; do WAE_TILE_WAVES=$W timeout -k 10 200 python dev/spmv_only.py C3 $R 2>&1 | tail -n 1 | sed "s/^/waves $W: /"; done; done
