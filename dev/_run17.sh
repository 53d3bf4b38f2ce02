cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_prof_newton
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/r3_prof_newton -o nw --output-format csv -- python3 dev/newton_c3.py C3 8 > gpurun_out/r3_prof_newton/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3_prof_newton/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# only the part after the Beyn pass: take the last 45% of dispatches by time? use kernel names with narrow grids instead: aggregate all, print top
import collections
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); t1 = int(rows[-1]["End_Timestamp"])
# find the start of the Newton phase: after the last beyn_accum_kernel
last = max(i for i, r in enumerate(rows) if "beyn_accum" in r["Kernel_Name"])
nw = rows[last + 1:]
agg = collections.defaultdict(lambda: [0, 0])
for r in nw:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:46]
    agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
span = (int(nw[-1]["End_Timestamp"]) - int(nw[0]["Start_Timestamp"])) / 1e9
print("newton phase: %.2f s wall span, %.2f s kernel time, %d launches" % (span, tot / 1e9, len(nw)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-46s %6d calls %8.1f ms avg %7.1f us %5.1f%%" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3, 100 * v[1] / tot))
PY
