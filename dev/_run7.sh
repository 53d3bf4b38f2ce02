cd $GRAFT_REPO_ROOT
WAE_SETUP_DEBUG=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-newton > gpurun_out/r3_first2.json 2> gpurun_out/r3_first2.err; grep "^.amg\|^.setup" gpurun_out/r3_first2.err
python - <<'P'
import json
j=json.loads(open('gpurun_out/r3_first2.json').read().strip().split('\n')[-1])
print(j['ms_per_step'], j['value_cold'], j['cold'], j['step_breakdown_seconds'])
P
