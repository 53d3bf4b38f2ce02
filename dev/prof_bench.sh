#!/bin/bash
# rocprofv3 kernel stats of a short bench run; prints the top kernels.  usage (GPU box): dev/prof_bench.sh OUTDIR [bench args]
set -e
OUT=${1:-gpurun_out/prof_bench}; shift || true
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT -o bench --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-newton "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel s %.3f" % (tot / 1e9))
for r in rows[:18]:
    print("%-44s %6s calls %9.1f ms avg %8.1f us %5.1f%%" % (r["Name"].split("(")[0].replace("void ", "")[:44], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
grep "^{" $OUT/log.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'iters', d['solver']['iters_total'])"
