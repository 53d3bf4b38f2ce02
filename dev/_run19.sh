cd $GRAFT_REPO_ROOT
export WAE_BENCH_BACKEND=gloo
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_b_n2.json 2> gpurun_out/r3_b_n2.err; echo rc=$?
tail -n 3 gpurun_out/r3_b_n2.err
python - <<'P'
import json
j=json.loads([l for l in open('gpurun_out/r3_b_n2.json') if l.startswith('{')][-1])
print(j['n_gpus'], j['value'], j['ms_per_step'], j['eigenpairs'], j['rank_gap'], j['eig_residual_max'], j['step_breakdown_seconds'], j['roofline']['frac'], j['config']['parallelism'][:80])
P
