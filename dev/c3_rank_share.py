#!/usr/bin/env python3
"""One rank's share of the C3 snapshot phase over G GPUs, timed on one GPU (what each of the G ranks does concurrently), for the two
ways of sharing it out (nlevp/distributed.py, csrc/mgpu.hip):
  columns: all S snapshot points for l/G probe columns, progressive (mode 0);
  hybrid : S/G snapshot points for all l columns from zero guesses (mode 3) + the basis of all S snapshots for l/G columns (mode 4;
           the other ranks' raw snapshots are stood in for by copies of the rank's own: same work).
usage: c3_rank_share.py [G ...]      output: one line per G, also gpurun_out/rank_share.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import wae_amd  # noqa: F401
from wae_amd.helmholtz.family import annulus_family
from wae_amd.nlevp import gauss_points
from wae_amd.nlevp.beyn import coefficient_table, snapshot_split, spread_order

preset = os.environ.get("PRESET", "C3")
L, pb = annulus_family(preset, tau=2e-4)
d = pb["d"]
L.solver_tol = 1e-10; L.solver_maxit = 400; L.solver_ref = 2 * np.pi * 500.0
L.solver_opts = {"batch": 64, "restart": 40, "sweeps": 1}
fam = L.ensure_solver()
G = np.array([150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]) * 2 * np.pi
N, l, S = (64, 8, 40) if preset == "C3" else (32, 16, 40)
zs, ws = gauss_points(G, N)
ct = coefficient_table(L, zs)
idx, rest = snapshot_split(len(zs), S)
idx = spread_order(idx)
V = np.asfortranarray(np.random.default_rng(7).standard_normal((d, l)) + 0j)
buf = torch.zeros(d * l * 2 * 2, dtype=torch.float64, device="cuda:0")
kw = dict(K=1, tol=1e-10, maxit=400, out_dev=buf.data_ptr())
rows = []
for world in [int(a) for a in sys.argv[1:]] or [8, 4, 2, 1]:
    ls = l // world
    local = torch.empty(S * d * ls * 2, dtype=torch.float64, device="cuda:0")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fam.beyn_moments_rb(zs[idx], ws[idx], ct[idx], V[:, :ls], 0, S, Q_dev=local.data_ptr(), l_total=l, col0=0, **kw)
        torch.cuda.synchronize(); t_col = time.perf_counter() - t0
    its_col = fam.last_info["iters_total"]
    row = {"G": world, "columns_s": t_col, "columns_column_iterations": its_col}
    if world > 1:
        Sb = (S // world) * world
        per = Sb // world
        mine = idx[0:Sb:world]
        raw = torch.empty(per * d * l * 2, dtype=torch.float64, device="cuda:0")
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fam.beyn_moments_rb(zs[mine], ws[mine], ct[mine], V, 3, per, Q_dev=raw.data_ptr(), **kw)
            torch.cuda.synchronize(); t_solve = time.perf_counter() - t0
        its_h = fam.last_info["iters_total"]
        sl = raw.view(per, d, world, ls, 2)[:, :, 0].contiguous()                       # own columns of own snapshots
        stand_in = sl.repeat(world, 1, 1, 1).contiguous()                                # S x d x ls (copies: same orthogonalisation work,
        stand_in += 1e-3 * torch.randn_like(stand_in)                                    #  made independent so that nothing is dropped)
        for rep in range(2):
            work = stand_in.clone()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fam.beyn_moments_rb(zs[idx[:Sb]], ws[idx[:Sb]], ct[idx[:Sb]], V[:, :ls], 4, Sb, slot0=Sb, Q_dev=work.data_ptr(), accumulate=True, l_total=l, col0=0, **kw)
            torch.cuda.synchronize(); t_build = time.perf_counter() - t0
        row.update({"hybrid_solves_s": t_solve, "hybrid_basis_s": t_build, "hybrid_s": t_solve + t_build, "hybrid_column_iterations": its_h,
                    "alltoall_bytes_per_rank": per * d * l * 16 * (world - 1) // world})
        del raw, sl, stand_in, work
    rows.append(row)
    print(json.dumps(row), flush=True)
    del local
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"preset": preset, "S": S, "l": l, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "rank_share.json"), "w"), indent=1)
