#!/usr/bin/env python3
"""Benchmark of the NLEVP hot path on MI355X:  eigenpairs/sec of a Beyn contour solve on the synthetic annular
combustor at 1M DoF -- the configuration BASELINE.json quotes its metric on ("1M-DoF Helmholtz NLEVP", configs[2]:
995 328 DoF, n·exp(-iωτ) flame term, Beyn N=64 per edge, l=8); it fits one GPU (about 77 GB of HBM).  BASELINE configs[1]
(200k DoF, N=32) is `--preset C2 --N 32 --l 16`.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one complete pass of the hot path: all quadrature points of the contour (4 edges x 64 Gauss-Legendre
nodes = 256 shifted systems x l=8 probe columns), each solved on the device by multigrid-GMRES whose operator
application is the fused multi-term CSR SpMV; moment accumulation in HBM.  By default (--rb -1 = the package's
automatic rule, 40 of the 256 points) snapshot points are solved first and every other system starts from the Galerkin
projection of its solution on them: same moments to the inner tolerance, same stopping test; --rb 0
solves every system from a zero guess.  N>1: every rank solves all snapshot points for its l/N probe columns, the
per-column bases are all-gathered, the remaining points are dealt round-robin and the partial moment tensors summed
with one RCCL all-reduce over xGMI.  Then QR+SVD of the moments on the GPU, small eigenproblem, position test and the
residual check of every eigenpair (on the device) on rank 0.  Inputs (all term matrices, the multigrid hierarchy) are
resident in HBM before the timed region.

The printed JSON line carries, besides the driver's contract fields,
  roofline     : the dominant kernel (spmv_kernel, the fused multi-term SpMV at the solver's batch width) --
                 algorithmic bytes (SURVEY.md 8d formula) / average launch duration measured with HIP events on
                 the library's own stream;
  cpu_baseline : the reference-shaped path (sparse LU + l solves per quadrature point, scipy SuperLU = the oracle)
                 timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GAMMA_HZ = [150 - 150j, 1000 - 150j, 1000 + 150j, 150 + 150j]   # contour in Hz (x 2π -> rad/s)
HBM_PEAK_GBS = 8000.0


def cpu_baseline(l, N, n_in, d_bench, tau, n_flame, budget_s=30.0):
    """Reference-shaped quadrature point on the host -- assemble L(z), sparse LU, l solves (beyn.jl:62-71; oracle/solvers.py,
    scipy SuperLU standing in for UMFPACK) -- timed on a ladder of annulus meshes of growing size with the SAME flame
    parameters as the GPU leg, until `budget_s` of CPU time is spent (at least two sizes).  The benchmark operator itself is
    out of reach of an in-run sample (one LU at 200k DoF: 579 s and 9.4 GB of fill, measured offline; at 1M DoF the fill no
    longer fits), so the figure for the benchmark size comes from the measured power law t_point ~ d^p and is labelled
    extrapolated.  One core: the port is serial like the reference (a single Julia thread + UMFPACK)."""
    from oracle import solvers as OS
    from wae_amd.helmholtz import annulus
    from threadpoolctl import threadpool_limits
    ladder = [(60, 30, 8), (72, 36, 8), (84, 42, 9), (96, 48, 10), (108, 54, 11)]
    z = 2 * np.pi * (575 + 150j)
    rows, spent = [], 0.0
    with threadpool_limits(limits=1):            # one core for real: SuperLU's BLAS calls would otherwise fan out over every host core
        OS._solve(annulus.build("tiny")["terms"]["M"].tocsc() + 0j, OS.initial_V(1152, 1))     # (loads SuperLU outside the timings)
        for grid in ladder:
            pb = annulus.build(grid=grid, n=n_flame, tau=tau)
            T = pb["terms"]
            t0 = time.time()
            A = (z * z * T["M"] + T["K"] + z * 1e15 * T["C"] + n_flame * np.exp(-1j * z * tau) * T["Q"]).tocsc()
            OS._solve(A, OS.initial_V(pb["d"], l))
            dt = time.time() - t0
            rows.append({"d": int(pb["d"]), "seconds_per_point": dt})
            spent += dt
            if len(rows) >= 2 and spent + 2.5 * dt > budget_s:
                break
    measured = None                  # one reference-shaped point at BASELINE configs[1] size, timed on the build container (dev/cpu_point_c2.py)
    mfile = os.path.join(ROOT, "profiles", "r03_cpu_point_C2.json")
    if os.path.exists(mfile):
        measured = json.load(open(mfile))
    p = float(np.polyfit(np.log([r["d"] for r in rows]), np.log([r["seconds_per_point"] for r in rows]), 1)[0])
    t_bench = rows[-1]["seconds_per_point"] * (d_bench / rows[-1]["d"]) ** p
    npts = 4 * N
    return {"value": n_in / (npts * t_bench), "unit": "eigenpairs/sec", "cores": 1, "kind": "port", "extrapolated": True,
            "host_cores": os.cpu_count(), "exponent": p, "seconds_per_point_at_benchmark_size": t_bench, "samples": rows,
            "tau": tau, "n": n_flame,
            "measured_point_C2": measured,
            "sample": f"1 quadrature point (assemble L(z) + SuperLU + {l} solves = the reference's per-point work, beyn.jl:62-71) on "
                      f"annulus meshes of {', '.join(str(r['d']) for r in rows)} DoF with the GPU leg's flame parameters, 1 of the host's "
                      f"{os.cpu_count()} cores; t_point ~ d^{p:.2f} extrapolated to d = {d_bench}: {t_bench:.0f} s per point, "
                      f"value = {n_in} eigenpairs / ({npts} points x t_point).  A lower bound on the CPU time: the fill exponent grows "
                      f"with d" + (f" (measured, profiles/r03_cpu_point_C2.json: {measured['seconds_per_point']:.0f} s and {measured['peak_rss_GB']:.1f} GB for ONE "
                                   f"point at {measured['d']} DoF, l = {measured['l']}, one core of the build container; the ladder's power law "
                                   f"gives {rows[-1]['seconds_per_point'] * (measured['d'] / rows[-1]['d']) ** p:.0f} s there)." if measured else ".")}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run as a CHILD process
    (this process never imports torch nor touches a GPU), pass its output through and return its exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["WAE_BENCH_SELF_LAUNCHED"] = "1"
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    # (torch.distributed.run's argparse resolves abbreviations over the WHOLE command line: a script option that is a prefix of one of
    # its own -- --l of --log-dir, --n of --nnodes -- is refused as ambiguous even behind the script name: pass the long spellings)
    long_names = {"--l": "--probe-columns", "--n": "--flame-n"}
    argv = []
    for a in sys.argv[1:]:
        head, eq, tail = a.partition("=")
        argv.append(long_names.get(head, head) + eq + tail)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    print(f"bench.py: launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:                       # rank 0's JSON line (and anything else the ranks print), as it comes
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--preset", default="C3")
    ap.add_argument("--l", "--probe-columns", dest="l", type=int, default=8)
    ap.add_argument("--N", type=int, default=64)
    ap.add_argument("--K", type=int, default=2,
                    help="moments 0..2K-1 (beyn.jl:50-52).  K = 2 makes the Hankel matrix l*K = 16 columns wide from the same 2 048 solves: "
                         "the 8 eigenvalues inside then show as a GAP in the singular values (rank_gap) instead of saturating an 8-column matrix")
    ap.add_argument("--tol", type=float, default=1e-10)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--restart", type=int, default=40)
    ap.add_argument("--sweeps", type=int, default=1)
    ap.add_argument("--jacw", type=float, default=0.8, help="Jacobi weight of the V-cycle's smoother")
    ap.add_argument("--n", "--flame-n", dest="n", type=float, default=1.0)
    ap.add_argument("--tau", type=float, default=2e-4)
    ap.add_argument("--rb", type=int, default=-1,
                    help="snapshot points for projected initial guesses (wae_beyn_moments_rb); -1 = the package's automatic "
                         "rule min(40, points/2); 0 = every system from a zero guess")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-newton", action="store_true", help="skip the Newton-type refinement figures")
    ap.add_argument("--cpu-budget", type=float, default=30.0, help="seconds of host time the CPU baseline may spend")
    ap.add_argument("--mgpu", action="store_true",
                    help="drive the N GPUs from ONE process through wae_beyn_moments_mgpu (the path a Julia host takes: one family handle "
                         "per device, a host thread + stream per device inside the library, RCCL by dlopen) instead of one rank per GPU")
    ap.add_argument("--rehearse", action="store_true",
                    help="rendezvous only (no GPU needed): every rank joins the process group, one all-reduce, rank 0 prints a JSON line "
                         "with the rank count -- what tests/ uses to check on a CPU that --gpus N really starts N ranks")
    args = ap.parse_args()
    if args.rb < 0:
        npts = 4 * args.N
        args.rb = min(40, npts // 2)      # nlevp/beyn.py compute_moment_matrices, automatic rule
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    # --gpus N means N ranks.  Started plainly (no WORLD_SIZE in the environment) this process only LAUNCHES them -- as a child
    # process, before torch is imported or the GPU touched -- relays rank 0's JSON line and leaves with the child's exit code.
    # Started by torch.distributed.run (the driver's way) the environment must agree with --gpus.
    env_world = os.environ.get("WORLD_SIZE")
    if args.mgpu:
        if env_world not in (None, "1"):
            sys.exit(f"bench.py --mgpu is a single-process mode; WORLD_SIZE={env_world} is set")
    elif env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    elif int(env_world or "1") != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: start it as `python bench.py --gpus N` (it launches its "
                 f"own ranks) or under torch.distributed.run with --nproc-per-node equal to --gpus")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = args.gpus if args.mgpu else 1          # devices driven by THIS process (--mgpu: all of them, one rank)
    if args.rehearse:
        tot = torch.ones(1, dtype=torch.float64)
        if world > 1:
            dist.init_process_group(os.environ.get("WAE_BENCH_BACKEND", "gloo"))
            dist.all_reduce(tot)
            dist.barrier()
        if rank == 0:
            print(json.dumps({"rehearsal": True, "n_gpus": world, "ranks_counted": int(tot.item()), "gpus_requested": args.gpus,
                              "launched_by": "bench.py" if os.environ.get("WAE_BENCH_SELF_LAUNCHED") else "external launcher"}))
        if world > 1:
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if args.mgpu and ndev > torch.cuda.device_count() and os.environ.get("WAE_MGPU_EXCHANGE") != "copy":
        sys.exit(f"bench.py --mgpu --gpus {ndev}: only {torch.cuda.device_count()} device(s) visible (several handles on one device "
                 f"are a rehearsal that needs WAE_MGPU_EXCHANGE=copy)")
    backend = os.environ.get("WAE_BENCH_BACKEND", "nccl")      # "gloo" only to rehearse N>1 on a one-GPU box
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    import wae_amd  # noqa: F401
    from wae_amd.helmholtz.family import annulus_family
    from wae_amd.nlevp import compute_moment_matrices, gauss_points, initialize_V, moments2eigs, pos_test
    from wae_amd.nlevp.distributed import (_guard, allreduce_sum_, beyn_moments_distributed_rb, fail_together, moments2eigs_device,
                                           shard_points)
    from wae_amd.nlevp.beyn import inpoly

    t0 = time.time()
    L, pb = annulus_family(args.preset, device=local, n=args.n, tau=args.tau)     # the input producer (`discretize`): numpy on the host
    d = pb["d"]
    t_build = time.time() - t0
    L.solver_tol = args.tol
    L.solver_maxit = 400
    L.solver_ref = 2 * np.pi * float(os.environ.get("WAE_REF_HZ", "500"))
    nshare = ndev if args.mgpu else world          # how many devices share the contour
    L.solver_opts = {"batch": args.batch, "restart": args.restart, "sweeps": args.sweeps, "jacobi_weight": args.jacw,
                     # workspace hints: the snapshot store of the passes to come is mapped during the set-up
                     "probe_columns": args.l // nshare if (nshare > 1 and args.l % nshare == 0) else args.l, "snapshots": args.rb}
    Ls = [L]                                       # --mgpu: one replica of the family per device, all driven from here
    for g in range(1, ndev):
        Lg = L.copy()
        Lg.device_id = g % torch.cuda.device_count()
        Ls.append(Lg)
    from concurrent.futures import ThreadPoolExecutor
    from wae_amd.nlevp.distributed import warm_up_dense_linalg
    warm_up_dense_linalg(torch.device("cuda", local), cols=args.l, K=args.K)     # (process-level library handles: not part of a solver call)
    t0 = time.time()
    with ThreadPoolExecutor(ndev) as ex:           # wae_family_create: conversion + upload of the term matrices (the library
        list(ex.map(lambda Lg: Lg.device(), Ls))   # releases the GIL: one host thread per device)
    torch.cuda.synchronize()
    t_upload = time.time() - t0
    # The benchmark boxes are restored micro-VMs: the first process to touch a page of guest memory pays the HOST's fault for it, and
    # the set-up's fresh vectors are such pages when this is the first large process on the box (set-up 1.4-2.5 s against 0.95-1.05 s
    # in every later process).  value_cold is reported AS MEASURED (no prefault); a second cold call on a fresh handle at the end of
    # the run (value_cold_second_handle) shows the same call with the host's pages already backed.  WAE_BENCH_PREFAULT_GB=<n> touches
    # and frees n GB before the clock starts (round 3's figure; off by default).
    t0 = time.time()
    prefault_gb = float(os.environ.get("WAE_BENCH_PREFAULT_GB", "0"))      # (round 4: off by default -- value_cold is what a first process sees)
    if prefault_gb > 0:
        blk = np.empty(int(prefault_gb * 2**30) // 8, dtype=np.float64)
        blk.reshape(-1, 512)[:, 0] = 0.0            # one write per 4-KB page
        del blk
    t_prefault = time.time() - t0
    t0 = time.time()
    with ThreadPoolExecutor(ndev) as ex:           # wae_solver_setup: multigrid hierarchy (part of the metric's "everything else")
        fams = list(ex.map(lambda Lg: Lg.ensure_solver(), Ls))
    fam = fams[0]
    for g in range(torch.cuda.device_count() if args.mgpu else 0):
        torch.cuda.synchronize(g)
    torch.cuda.synchronize()
    t_setup = time.time() - t0

    G = np.array(GAMMA_HZ) * 2 * np.pi
    zs, ws = gauss_points(G, args.N)
    zr, wr = shard_points(zs, ws, rank, world)           # round-robin shard of the quadrature points
    # probe matrix (beyn.jl:43, random=true), seeded; column-major like the Julia array the reference would hand over
    V = np.asfortranarray(np.random.default_rng(7).standard_normal((d, args.l)) + 0j)
    K = args.K
    # K > 1: moments in z' = (z - centre)/radius of the contour (same systems, better-conditioned Hankel matrix; see distributed.py)
    zmap = (complex(np.mean(G)), float(np.max(np.abs(G - np.mean(G))))) if (K > 1 and args.rb > 0) else None
    buf = torch.zeros(d * args.l * 2 * K * 2, dtype=torch.float64, device=f"cuda:{local}")

    tim = {}

    def step():
        nonlocal buf
        t = [time.time()]
        if args.mgpu:
            # ONE library call drives every device (wae_beyn_moments_mgpu): snapshot phase by probe column or by point, RCCL all-gather
            # of the bases, projected phase round-robin, RCCL reduce to device 0, moments to the host array the ABI returns them in;
            # the tail below wants them in HBM again (a Julia host would run its own dense tail on that array).
            from wae_amd.nlevp.distributed import beyn_moments_mgpu
            A, info = beyn_moments_mgpu(Ls, G, V, K=K, N=args.N, nsnap=args.rb, zmap=zmap)
            t.append(time.time())
            buf.copy_(torch.from_numpy(A.ravel(order="K").view(np.float64)))
            torch.cuda.synchronize()
            t.append(time.time())
            info = dict(info)
        elif args.rb > 0:
            # snapshot points -> all-gather of the snapshot store -> projected initial guesses -> all-reduce of the moments
            ph = {}
            buf, info = beyn_moments_distributed_rb(L, G, V, K, args.N, args.rb, timings=ph, zmap=zmap)
            t.append(t[0] + ph["snapshots"] + ph["allgather"] + ph["projected"])
            t.append(time.time())
            if rank == 0:
                tim.update({"snapshot_solves": ph["snapshots"], "allgather": ph["allgather"], "projected_solves": ph["projected"]})
        else:
            _, err = _guard(compute_moment_matrices, L, G, V, K=K, N=args.N, points=(zr, wr), out_dev=buf.data_ptr(), rb=0)
            fail_together(err, "beyn moments")                # (a rank whose share failed must not leave the others in the all-reduce)
            t.append(time.time())
            info = dict(fam.last_info)
            allreduce_sum_(buf)                              # sum of the partial moment tensors (RCCL over xGMI)
            torch.cuda.synchronize()
            t.append(time.time())
        res = None
        if rank == 0:
            t.append(time.time())
            # SVD on the GPU, small eig on the host; K > 1: through the Gram matrix (the Hankel matrix is rank-deficient by construction:
            # Householder QR of the 2d x 16 matrix alone took 0.12 s of a pass), directions above 1e-6 sigma_1 kept (beyn.jl:92-95 `tol`)
            Om, Pd, S = moments2eigs_device(buf, (d, args.l, 2 * K), gram_rel_tol=1e-6 if K > 1 else 0.0)
            if zmap is not None:
                Om = zmap[0] + zmap[1] * Om                                  # back from the mapped variable of the moments
            mask = np.array([inpoly(w, G) for w in Om], dtype=bool)         # pos_test (beyn.jl:104-107)
            Om = Om[mask]
            Pt = Pd[:, torch.from_numpy(mask).to(Pd.device)].T.contiguous()   # (n, d) row-major = column-major d x n, in HBM
            torch.cuda.synchronize()
            t.append(time.time())
            r = (fam.eig_residuals(np.array([L.coefficients(w) for w in Om]), P_dev=Pt.data_ptr())
                 if len(Om) else np.zeros(0))                                # the eigenvectors never leave the device
            t.append(time.time())
            res = (Om, r, S, Pd, mask)
            for name, a, b in (("moments", 0, 1), ("allreduce", 1, 2), ("d2h", 2, 3), ("svd_eig", 3, 4), ("residuals", 4, 5)):
                tim[name] = t[b] - t[a]
            if args.mgpu:
                tim["moments_h2d_for_tail"] = tim.pop("allreduce")      # (the ABI returned them on the host)
        return info, res

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the first pass is timed on its own: with the solver set-up it is the cost of ONE cold solver call, the metric as SURVEY
    # 8d words it ("upload excluded, everything else included"); the K timed steps below amortise the set-up
    sync()
    first, t_first, tim_first = None, None, {}
    if args.warmup >= 1:                          # (the first of the W untimed warm-up steps)
        t0 = time.time()
        first = step()
        sync()
        t_first = time.time() - t0
        tim_first = dict(tim)
    for _ in range(args.warmup - 1):
        step()
    sync()
    t0 = time.time()
    last = None
    for _ in range(args.steps):
        last = step()
    sync()
    dt = torch.tensor([time.time() - t0], dtype=torch.float64, device=f"cuda:{local}")
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())

    if rank == 0:
        info, (Om, r, S, Pd, inside_mask) = last
        good = r <= 1e-6
        n_eig = int(good.sum())
        good_mask_full = np.zeros(len(inside_mask), dtype=bool)        # columns of Pd (all Ritz pairs) that are verified eigenpairs
        good_mask_full[np.nonzero(inside_mask)[0][good]] = True
        # roofline of the dominant kernel, measured live (HIP events on the library's stream)
        cz = L.coefficients(2 * np.pi * (500 + 20j))
        mask = [1 if c != 0 else 0 for c in cz]
        rb = args.batch
        ms = fam.bench_spmv(cz, r=rb, reps=50)
        abytes = fam.spmv_bytes(r=rb, mask=mask)
        ms1 = fam.bench_spmv(cz, r=1, reps=50)
        ms8 = fam.bench_spmv(cz, r=8, reps=50)
        import ctypes as _C
        from wae_amd import _lib as _wl
        tri = _C.c_double(0.0)                 # device triad a = b + s*c over 2^27 doubles: the streaming rate this GPU attains
        _wl.check(_wl.lib().wae_bench_triad(int(os.environ.get("LOCAL_RANK", 0)), 1 << 27, 20, _C.byref(tri)))
        traffic = None       # HBM bytes per launch from the PMC passes committed under profiles/ (not collectable in-run)
        tfile = next((f for f in (os.path.join(ROOT, "profiles", f"r0{k}_spmv_traffic_{args.preset}.json") for k in (4, 3, 2)) if os.path.exists(f)),
                     os.path.join(ROOT, "profiles", f"r04_spmv_traffic_{args.preset}.json"))
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            if tj.get("preset") == args.preset and tj.get("r") == rb:
                traffic = tj["traffic_bytes"]
        tiled = os.environ.get("WAE_SPMV_TILE", "1") != "0" and os.environ.get("WAE_REORDER", "1") != "0"
        roof = {"bound": "hbm", "kernel": ("spmv_tile_kernel<true, 2, 2> (+ spmv_side_kernel: the rows of the boundary / flame terms, ~5 % of the time)"
                                           if tiled else "spmv_lds_kernel<4>")
                                          + ": one fused multi-term complex CSR operator product of the fine level, r columns",
                "achieved": abytes / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": abytes / ms / 1e6 / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": os.path.relpath(tfile, ROOT) if traffic is not None else None,
                "r": rb, "us_per_launch": ms * 1e3, "algorithmic_bytes": int(abytes),
                "triad_GBps": tri.value, "frac_of_triad": abytes / ms / 1e6 / tri.value if tri.value else None,
                # narrow products: EFFECTIVE rates (algorithmic bytes / time).  Repeated launches re-read the matrix from the L2 /
                # Infinity Cache, so these are not HBM rates and carry no fraction of the HBM peak.
                "r1": {"us": ms1 * 1e3, "effective_GB/s": fam.spmv_bytes(r=1, mask=mask) / ms1 / 1e6, "note": "algorithmic bytes / time; cache-resident matrix, not an HBM rate"},
                "r8": {"us": ms8 * 1e3, "effective_GB/s": fam.spmv_bytes(r=8, mask=mask) / ms8 / 1e6, "note": "algorithmic bytes / time; cache-resident matrix, not an HBM rate"}}
        # the 4-lane instantiation of the tile kernel: the level-1 operator and the fine-to-coarse restriction (13-15 % of the kernel time)
        try:
            msl, bl = fam.bench_spmv_level(cz, which=0, level=1, r=rb, reps=50)
            msr, br = fam.bench_spmv_level(cz, which=1, level=0, r=rb, reps=50)
            roof["level1"] = {"kernel": "spmv_tile_kernel<true, 4, 2>: level-1 operator product, r columns", "us_per_launch": msl * 1e3,
                              "algorithmic_bytes": int(bl), "achieved": bl / msl / 1e6, "frac": bl / msl / 1e6 / HBM_PEAK_GBS, "unit": "GB/s"}
            roof["restriction"] = {"kernel": "spmv_tile_kernel<true, 4, 2> (unit coefficients): restriction level 0 -> 1, r columns",
                                   "us_per_launch": msr * 1e3, "algorithmic_bytes": int(br), "achieved": br / msr / 1e6,
                                   "frac": br / msr / 1e6 / HBM_PEAK_GBS, "unit": "GB/s"}
            msp, bp = fam.bench_spmv_level(cz, which=2, level=0, r=rb, reps=50)
            roof["prolongation"] = {"kernel": "prolong_tiles_kernel: x += P e, level 1 -> 0, r columns (in place: the fine vector is read and written)",
                                    "us_per_launch": msp * 1e3, "algorithmic_bytes": int(bp), "achieved": bp / msp / 1e6,
                                    "frac": bp / msp / 1e6 / HBM_PEAK_GBS, "unit": "GB/s"}
        except Exception as e:          # noqa: BLE001  (a hierarchy without a tiled level 1)
            roof["level1"] = {"error": str(e)}
        out = {
            "metric": "eigenpairs/sec", "value": n_eig * args.steps / dt, "unit": "eigenpairs/sec",
            "n_gpus": world * ndev, "rccl_ranks": (dist.get_world_size() if world > 1 else 1) if not args.mgpu else ndev,
            "launch": ("one process, wae_beyn_moments_mgpu over %d device handle(s)" % ndev) if args.mgpu else
                      ("%d rank(s), one process per GPU (torch.distributed, backend %s)" % (world, backend)),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"annular combustor Helmholtz NLEVP (P1), preset {args.preset}: d={d}, "
                                   f"L(w)=w^2 M+K+w Y C+n exp(-i w tau) Q, Beyn l={args.l} K={K} N={args.N}/edge "
                                   f"({4 * args.N} shifted systems x {args.l} columns), contour 150..1000 Hz x +-150 Hz, inner tol {args.tol:g}",
                       "parallelism": ("single host process: wae_beyn_moments_mgpu, a host thread + stream per device, %d device(s); "
                                       "snapshot phase by probe column (by point when l is not divisible), bases all-gathered, projected "
                                       "points round-robin, moments reduced to device 0 and returned in host memory" % ndev) if args.mgpu else
                                      ((f"{args.rb} snapshot points solved first, split by probe column over {world} GPU(s) "
                                        f"({args.l // world} columns each), their solutions all-gathered; " if args.l % world == 0 else
                                        f"{args.rb} snapshot points solved first, round-robin over {world} GPU(s), their solutions all-gathered; ")
                                       + f"the other {4 * args.N - args.rb} points round-robin over the GPUs, starting from the projection on "
                                       "the snapshots; one all-reduce of the moments"
                                       if args.rb > 0 else
                                       f"quadrature points round-robin over {world} GPU(s), one all-reduce of the moments"),
                       "batch_columns": args.batch},
            "eigenpairs": n_eig, "eigenvalues_hz": [[float(x.real), float(x.imag)] for x in np.sort_complex(Om[good] / 2 / np.pi)],
            "eig_residual_max": float(r[good].max()) if n_eig else None, "n_inside_before_residual_test": int(len(Om)),
            "singular_values": [float(s) for s in S],
            # the count certifies itself: singular value number n_inside against the next one (Hankel matrix d*K x l*K, beyn.jl:85-95)
            "rank_gap": (float(S[n_eig - 1] / S[n_eig]) if 0 < n_eig < len(S) else None),
            "value_cold": int((first[1][1] <= 1e-6).sum()) / (t_setup + t_first) if first is not None else None,
            "cold": {"solver_setup_seconds": t_setup, "first_pass_seconds": t_first, "upload_seconds": t_upload,
                     "problem_build_seconds": t_build, "first_pass_breakdown_seconds": {k: round(v, 4) for k, v in tim_first.items()},
                     "host_prefault": {"GB": prefault_gb, "seconds": round(t_prefault, 3),
                                       "what": "host memory touched and freed before the solver's clock (restored micro-VM: first touch of guest memory)"},
                     "note": "value_cold = eigenpairs / (wae_solver_setup + first Beyn pass): one cold solver call with the term "
                             "matrices already uploaded; value = the same pass with the hierarchy resident (K timed steps)"},
            "solver": {**info, "setup_seconds": t_setup},
            "step_breakdown_seconds": {k: round(v, 4) for k, v in tim.items()},
            "roofline": roof,
        }
        if world == 1 and not args.no_newton:
            # The Newton-type half of the hot path (north_star names it beside Beyn): every Beyn estimate refined by `householder`
            # (Householder.jl:70-192; two shift-invert Arnoldi processes + one first-order perturbation per Newton step), all
            # estimates in one lock-step batch.  Outside the timed region; its own figures.
            from wae_amd.nlevp import householder_many
            P_host = Pd[:, torch.from_numpy(good_mask_full).to(Pd.device)].cpu().numpy() if n_eig else np.zeros((d, 0), dtype=complex)
            t0n = time.time()
            tol_beyn = L.solver_tol
            L.solver_tol = 1e-12                  # the Ritz test of the shift-invert Arnoldi (1e-12) needs inner solves at least as accurate
            try:
                nstats = {}
                outs = householder_many(L, list(Om[good]), maxiter=6, tol=1e-8 * 2 * np.pi, v0s=P_host, stats=nstats) if n_eig else []
            finally:
                L.solver_tol = tol_beyn
            t_newton = time.time() - t0n
            conv = [o for o in outs if o[2] in (0, 1)]
            shift = [abs(o[0].params["ω"] - w) / abs(w) for o, w in zip(outs, Om[good])]
            out["newton"] = {"what": "householder_many over the Beyn estimates (tol 1e-8 Hz-relative step, at most 6 Newton steps)",
                             "eigenpairs_refined_per_sec": len(conv) / t_newton if t_newton > 0 else None, "seconds": t_newton,
                             "refined": len(conv), "of": len(outs), "newton_steps": [int(o[1]) for o in outs],
                             "largest_relative_shift_from_beyn_estimate": float(max(shift)) if shift else None,
                             "phases": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in nstats.items()},
                             "spmv_r1": roof["r1"], "spmv_r8": roof["r8"]}
        if world == 1 and not args.mgpu and first is not None and not os.environ.get("WAE_BENCH_NO_SECOND_COLD"):
            # a second cold solver call (fresh handle: wae_family_create outside the clock, then wae_solver_setup + one pass) now that the
            # host's memory has been touched by the first: what value_cold is when the process is not the first on its box
            for Lg in Ls:
                Lg._drop_device()
            L2 = L.copy()
            L2.device()
            torch.cuda.synchronize()
            t0c = time.time()
            fam2 = L2.ensure_solver()
            torch.cuda.synchronize()
            t_setup2 = time.time() - t0c
            buf2, _ = beyn_moments_distributed_rb(L2, G, V, K, args.N, args.rb, zmap=zmap) if args.rb > 0 else (None, None)
            n2 = None
            if buf2 is not None:
                Om2, Pd2, _ = moments2eigs_device(buf2, (d, args.l, 2 * K), gram_rel_tol=1e-6 if K > 1 else 0.0)
                if zmap is not None:
                    Om2 = zmap[0] + zmap[1] * Om2
                m2 = np.array([inpoly(w, G) for w in Om2], dtype=bool)
                Pt2 = Pd2[:, torch.from_numpy(m2).to(Pd2.device)].T.contiguous()
                r2 = fam2.eig_residuals(np.array([L2.coefficients(w) for w in Om2[m2]]), P_dev=Pt2.data_ptr()) if m2.any() else np.zeros(0)
                n2 = int((r2 <= 1e-6).sum())
            torch.cuda.synchronize()
            t_call2 = time.time() - t0c
            out["value_cold_second_handle"] = (n2 / t_call2) if n2 else None
            out["cold"]["second_handle"] = {"solver_setup_seconds": t_setup2, "call_seconds": t_call2, "eigenpairs": n2}
            L2._drop_device()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.l, args.N, max(n_eig, 1), d, args.tau, args.n, budget_s=args.cpu_budget)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
