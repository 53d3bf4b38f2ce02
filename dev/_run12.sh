cd $GRAFT_REPO_ROOT
for W in "0.8 0.8" "0.9 0.7" "0.7 0.9" "1.0 0.6" "0.6 1.0" "0.65 1.2" "0.9 0.9" "0.7 0.7"; do
  set -- $W
  WAE_JAC_POST=$2 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-newton --jacw $1 > gpurun_out/r3_jw.json 2> gpurun_out/r3_jw.err
  python - "$W" <<'P'
import json,sys
try:
    j=json.loads(open('gpurun_out/r3_jw.json').read().strip().split('\n')[-1])
    print(sys.argv[1], round(j['ms_per_step'],1), j['solver']['iters_total'], j['eigenpairs'], j['solver']['n_unconverged'], '%.2e'%j['eig_residual_max'])
except Exception as e: print(sys.argv[1], 'failed', open('gpurun_out/r3_jw.err').read()[-300:])
P
done
