#!/bin/bash
# rocprofv3 evidence for the fine-level SpMV at r = 64 (run on the GPU box):  dev/collect_spmv_profile.sh PRESET TAG
# kernel-trace stats and two separate PMC passes (FETCH_SIZE, WRITE_SIZE), written under gpurun_out/prof_TAG/
set -e
P=${1:-C3}; TAG=${2:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 dev/spmv_only.py $P 64 > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch --output-format csv -- python3 dev/spmv_only.py $P 64 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write --output-format csv -- python3 dev/spmv_only.py $P 64 > $OUT/write.log 2>&1
if [ "$P" = C3 ]; then   # wave-level and LDS counters of the same launches (their own passes)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/sq -o sq --output-format csv -- python3 dev/spmv_only.py $P 64 > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/lds -o lds --output-format csv -- python3 dev/spmv_only.py $P 64 > $OUT/lds.log 2>&1
# (round 4) L2 requests: how many, how many reads, how many hit
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/l2hit -o l2hit --output-format csv -- python3 dev/spmv_only.py $P 64 > $OUT/l2hit.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum --kernel-trace -d $OUT/l2req -o l2req --output-format csv -- python3 dev/spmv_only.py $P 64 > $OUT/l2req.log 2>&1
fi
python3 - "$OUT" "$P" <<'PY'
import csv, glob, json, sys, collections
out, preset = sys.argv[1], sys.argv[2]
res = {"preset": preset, "r": 64}
for name in ("fetch", "write"):
    f = glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True)[0]
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "spmv" in r["Kernel_Name"]:
            vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        res[f"{name}:{k}"] = {"dispatches": len(v), "mean_after_3": sum(v[3:]) / max(1, len(v[3:]))}
f = glob.glob(f"{out}/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "spmv" in r["Name"]:
        res["stats:" + r["Name"].split("(")[0]] = {k: r[k] for k in ("Calls", "AverageNs", "MinNs", "MaxNs")}
print(json.dumps(res, indent=1))
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
PY
