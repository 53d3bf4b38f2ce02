cd $GRAFT_REPO_ROOT
WAE_SETUP_DEBUG=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-newton > gpurun_out/r3_setup3.json 2> gpurun_out/r3_setup3.err
grep "^\[setup\]\|^\[amg\]\|^\[create\]" gpurun_out/r3_setup3.err
python - <<'P'
import json
j=json.loads(open('gpurun_out/r3_setup3.json').read().strip().split('\n')[-1])
print(j['ms_per_step'], j['value_cold'], j['cold'], j['solver']['iters_total'], j['eig_residual_max'])
P
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "solve or tiles or beyn or G1 or G5" 2>&1 | tail -n 3
