cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-budget 10 > gpurun_out/r3_b2.json 2> gpurun_out/r3_b2.err; echo "rc=$?"
python - <<'P'
import json
j=json.loads(open('gpurun_out/r3_b2.json').read().strip().split('\n')[-1])
print(j['value'], j['ms_per_step'], j['value_cold'], j['eigenpairs'], j['rank_gap'], j['eig_residual_max'])
print(j['singular_values'])
print(j['roofline'])
print(j['newton'])
print({k:v for k,v in j['cpu_baseline'].items() if k not in ('samples',)})
P
