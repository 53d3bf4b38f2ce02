#!/bin/bash
# L2 hit / miss and fabric request counters of the fine-level operator product (run on the GPU box): dev/probes/l2_counters.sh PRESET TAG
set -e
P=${1:-C3}; TAG=${2:-r04}
OUT=gpurun_out/l2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/hit -o hit --output-format csv -- python3 dev/spmv_only.py $P 64 > $OUT/hit.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d $OUT/ea -o ea --output-format csv -- python3 dev/spmv_only.py $P 64 > $OUT/ea.log 2>&1 || true
rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum --kernel-trace -d $OUT/req -o req --output-format csv -- python3 dev/spmv_only.py $P 64 > $OUT/req.log 2>&1 || true
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "spmv" in r["Kernel_Name"]:
            vals[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(vals.items()):
        print(k, len(v), sum(v[3:]) / max(1, len(v[3:])))
PY
