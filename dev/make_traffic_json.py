#!/usr/bin/env python3
"""profiles/<TAG>_spmv_traffic_<PRESET>.json and the csv copies from gpurun_out/prof_<TAG>_<PRESET>/ (dev/collect_spmv_profile.sh)."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ALG = {"C3": 2643060688, "C2": None}
for preset in ("C3", "C2"):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{preset}")
    if not os.path.isdir(src):
        continue
    S = json.load(open(os.path.join(src, "summary.json")))
    fetch = {k.split(":", 1)[1]: v["mean_after_3"] for k, v in S.items() if k.startswith("fetch:")}
    write = {k.split(":", 1)[1]: v["mean_after_3"] for k, v in S.items() if k.startswith("write:")}
    stats = {k.split(":", 1)[1]: v for k, v in S.items() if k.startswith("stats:")}
    kernels = sorted(fetch)
    traffic = sum(2.0 * fetch[k] + write[k] for k in kernels) * 1024.0
    out = {"preset": preset, "r": 64,
           "kernels": {k: {"FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write[k],
                           "rocprofv3_average_ns": float(stats[k]["AverageNs"]) if k in stats else None,
                           "rocprofv3_min_ns": float(stats[k]["MinNs"]) if k in stats else None} for k in kernels},
           "correction": "gfx950: FETCH_SIZE x2 for wide reads (MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported",
           "traffic_bytes": int(traffic),
           "source": f"profiles/{tag}_spmv_pmc_{preset}_FETCH_SIZE.csv, profiles/{tag}_spmv_pmc_{preset}_WRITE_SIZE.csv (separate rocprofv3 --pmc passes of "
                     f"dev/spmv_only.py {preset} 64 via dev/collect_spmv_profile.sh; first 3 of 53 dispatches discarded); one fine-level operator "
                     "product = spmv_side_kernel (rows of the boundary / flame terms) + spmv_tile_kernel"}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_spmv_traffic_{preset}.json"), "w"), indent=1)
    for name, dst in (("fetch/fetch_counter_collection.csv", f"{tag}_spmv_pmc_{preset}_FETCH_SIZE.csv"),
                      ("write/write_counter_collection.csv", f"{tag}_spmv_pmc_{preset}_WRITE_SIZE.csv"),
                      ("stats/stats_kernel_stats.csv", f"{tag}_spmv_only_{preset}_kernel_stats.csv"),
                      ("sq/sq_counter_collection.csv", f"{tag}_spmv_pmc_{preset}_SQ.csv"), ("lds/lds_counter_collection.csv", f"{tag}_spmv_pmc_{preset}_LDS.csv"),
                      ("l2hit/l2hit_counter_collection.csv", f"{tag}_spmv_pmc_{preset}_L2HIT.csv"), ("l2req/l2req_counter_collection.csv", f"{tag}_spmv_pmc_{preset}_L2REQ.csv")):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(ROOT, "profiles", dst))
    print(preset, json.dumps(out["kernels"], indent=1), "traffic GB", traffic / 1e9)
