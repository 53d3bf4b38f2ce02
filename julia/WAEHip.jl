# WAEHip.jl -- the reference-side binding of libwaehip.so (include/waehip.h).
#
# Host code stays Julia: this file is what a maintainer of WavesAndEigenvalues.jl adds (e.g. as
# src/NLEVP/WAEHip.jl, `include`d from src/NLEVP.jl) to run the NLEVP hot path on an MI355X.  It is a thin
# `ccall` layer: no CUDA.jl/AMDGPU.jl, no second code path.  It cannot be executed in the build container of this
# repository (no julia binary there); the same ABI is exercised from Python by tests/ (ctypes).
#
# Usage (drop-in for the hot path):
#     using WavesAndEigenvalues.NLEVP, .WAEHip
#     L  = discretize(mesh, dscrp, c)                 # unchanged (src/Helmholtz.jl:54)
#     Ld = WAEHip.DeviceFamily(L)                     # terms -> HBM once
#     Ω, P = WAEHip.beyn(Ld, Γ; l=16, N=32)           # same keywords as NLEVP.beyn (src/NLEVP/beyn.jl:34)
#     A = Ld(z); y = A*x; x = A\b; x = A'\b           # operator view instead of a SparseMatrixCSC
#     sol, n, flag = WAEHip.householder(Ld, Ω[1]; tol=1e-11)      # Householder.jl:70, same keywords / flags
#     sol, n, flag = WAEHip.mslp(Ld, Ω[1]; tol=1e-11)             # iterative_solvers.jl:93
#     WAEHip.perturb_fast!(sol, Ld, :τ, 30)                       # LinOpFam.jl:575; no multi-index files needed
#     res = WAEHip.eig_residuals(Ld, Ω, P)                        # which Ritz pairs are eigenpairs
#     Lds = [WAEHip.DeviceFamily(L; device=g) for g in 0:7]       # 8 GPUs from one Julia process:
#     A = WAEHip.compute_moment_matrices(Lds, Γ, V; N=64)         #   RCCL over xGMI inside the library
module WAEHip

using LinearAlgebra, SparseArrays
import FastGaussQuadrature
import ..NLEVP: LinearOperatorFamily, Term, inpoly

const libwaehip = get(ENV, "WAEHIP_LIB", "libwaehip.so")

struct SolveInfo
    iters_max::Int32; iters_total::Int32; n_unconverged::Int32; levels::Int32
    relres_max::Float64; seconds::Float64
end

const OP_N, OP_T, OP_C = Int32(0), Int32(1), Int32(2)

function check(code::Integer)
    if code < 0
        msg = unsafe_string(ccall((:wae_last_error, libwaehip), Cstring, ()))
        # map onto the exceptions the reference's solvers already catch (iterative_solvers.jl:192-210)
        code == -2 && throw(LinearAlgebra.SingularException(0))
        error("libwaehip error $code: $msg")
    end
    return code            # > 0: warning (max. iterations / stagnation of an inner solve)
end

"the reference's `\\` is a direct LU that solves or throws: an inner solve that stopped short of its tolerance must not pass
silently.  `fatal=true` (contour integrals: a stalled quadrature point corrupts every eigenpair) throws, otherwise a warning;
`quiet=true` is for callers that judge the outcome themselves (the Newton-type solvers on numerically singular operators)."
function report(code::Integer, info::SolveInfo, what::AbstractString; fatal::Bool=false, quiet::Bool=false)
    if info.n_unconverged > 0 && !quiet
        msg = "$what: $(info.n_unconverged) inner solve(s) did not reach the tolerance (largest relative residual $(info.relres_max), " *
              (code == 2 ? "stagnation" : "iteration limit") * ")"
        fatal ? error(msg) : @warn msg
    end
    return code
end

mutable struct DeviceFamily
    L::LinearOperatorFamily
    handle::Ptr{Cvoid}
    solver_ready::Bool
    tol::Float64
    maxit::Int32
    # symmetry_tol: opts[0] of wae_family_create_opts.  0 (default): `A'` is exactly `A'` -- a term is applied un-transposed for
    # A'*y / A'\\b only if it is bitwise symmetric.  Families from `discretize` (M, K, C symmetric by construction, assembled in floating
    # point) should pass 1e-14: their adjoint products then take the forward path (include/waehip.h).
    function DeviceFamily(L::LinearOperatorFamily; device::Integer=0, tol=1e-12, maxit=400, symmetry_tol::Float64=0.0)
        T = length(L.terms)
        d = size(L.terms[1].coeff, 1)
        # Helmholtz terms are SparseMatrixCSC{ComplexF64,UInt32} (Helmholtz.jl:407-408,515): pass colptr/rowval/nzval as they are
        mats = [SparseMatrixCSC{ComplexF64,UInt32}(sparse(t.coeff)) for t in L.terms]
        ptrs = [pointer(m.colptr) for m in mats]; idxs = [pointer(m.rowval) for m in mats]; vals = [pointer(m.nzval) for m in mats]
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve mats ptrs idxs vals begin
            if symmetry_tol == 0.0
                check(ccall((:wae_family_create, libwaehip), Cint,
                            (Ref{Ptr{Cvoid}}, Int64, Int32, Int32, Int32, Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Int32),
                            h, d, T, 4, 1, 0 #=WAE_CSC=#, ptrs, idxs, vals, device))
            else
                opts = Float64[symmetry_tol]
                check(ccall((:wae_family_create_opts, libwaehip), Cint,
                            (Ref{Ptr{Cvoid}}, Int64, Int32, Int32, Int32, Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Int32, Ptr{Float64}, Int32),
                            h, d, T, 4, 1, 0 #=WAE_CSC=#, ptrs, idxs, vals, device, opts, length(opts)))
            end
        end
        fam = new(L, h[], false, tol, maxit)
        finalizer(f -> (f.handle != C_NULL && ccall((:wae_family_destroy, libwaehip), Cint, (Ptr{Cvoid},), f.handle); f.handle = C_NULL), fam)
        return fam
    end
end

"scalar half of the functor `L(args...)` (LinOpFam.jl:482-526): one coefficient per term, 0 where the term is skipped"
function coefficients(L::LinearOperatorFamily, args...)
    if L.mode == :all
        for (var, val) in zip(L.active, args); L.params[var] = val; end
    end
    derivs = (L.mode == :all && length(args) == length(L.active)) ? zeros(Int, length(L.active)) : collect(args[end-length(L.active)+1:end])
    dd = Dict(zip(L.active, derivs))
    c = zeros(ComplexF64, length(L.terms))
    for (k, term) in enumerate(L.terms)
        (L.mode != :householder && term.operator == "__aux__") && continue
        any(d > 0 && !(var in term.varlist) for (var, d) in zip(L.active, derivs)) && continue
        ck = one(ComplexF64)
        for (func, pars) in zip(term.func, term.params)
            ck *= func([L.params[p] for p in pars]..., [get(dd, p, 0) for p in pars]...)
        end
        c[k] = ck
    end
    if L.mode in (:compact, :householder)
        c ./= prod(factorial.(float.(args[end-length(L.active)+1:end])))
    end
    return c
end

"what `L(z)` returns on the device: (family, coefficients, op) -- supports *, \\, ', size"
struct Operator
    fam::DeviceFamily
    c::Vector{ComplexF64}
    op::Int32
end
(fam::DeviceFamily)(args...) = Operator(fam, coefficients(fam.L, args...), OP_N)
Base.adjoint(A::Operator) = Operator(A.fam, A.c, A.op == OP_C ? OP_N : OP_C)
Base.size(A::Operator) = (n = size(A.fam.L.terms[1].coeff, 1); (n, n))

function Base.:*(A::Operator, X::StridedVecOrMat{ComplexF64})
    Y = similar(X)
    check(ccall((:wae_spmv_sum, libwaehip), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32, Int32),
                A.fam.handle, A.c, X, Y, size(X, 2), A.op))
    return Y
end

"number of HIP devices the library sees (wae_device_count) and its build string (wae_version)"
function device_count()
    n = Ref{Cint}(0)
    check(ccall((:wae_device_count, libwaehip), Cint, (Ref{Cint},), n))
    return Int(n[])
end
version() = unsafe_string(ccall((:wae_version, libwaehip), Cstring, ()))

"(d, T, nnz) of the resident family (wae_family_info)"
function family_info(fam::DeviceFamily)
    d = Ref{Int64}(0); T = Ref{Int32}(0); nz = Ref{Int64}(0)
    check(ccall((:wae_family_info, libwaehip), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int32}, Ref{Int64}), fam.handle, d, T, nz))
    return Int(d[]), Int(T[]), Int(nz[])
end

"algorithmic bytes of one `L(z)*X` with r columns over the terms flagged in `mask` (wae_family_spmv_bytes; SURVEY 8d formula)"
function spmv_bytes(fam::DeviceFamily; r::Integer=1, mask=nothing)
    m = mask === nothing ? C_NULL : UInt8.(mask .!= 0)
    return ccall((:wae_family_spmv_bytes, libwaehip), Int64, (Ptr{Cvoid}, Ptr{UInt8}, Int32), fam.handle, m, r)
end

"Y[:,j] = Σ_k C[k,j] op(A_k) X[:,j]: one coefficient column per column of X (wae_spmv_sum_cols) -- e.g. `L(ω_j)*v_j` or
`L(ω_j,1)*v_j` for all start values of `householder_many` in one launch.  C is T x r (column j = `coefficients(L, ω_j)`)."
function spmv_cols(fam::DeviceFamily, C::Matrix{ComplexF64}, X::Matrix{ComplexF64}; op::Int32=OP_N)
    size(C, 2) == size(X, 2) || error("spmv_cols: one coefficient column per column of X")
    Y = similar(X)
    check(ccall((:wae_spmv_sum_cols, libwaehip), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Int32, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32, Int32),
                fam.handle, C, size(C, 2), X, Y, size(X, 2), op))
    return Y
end

"y = Σ_k c[k] A_k X[:,k]: one input column per term (wae_spmv_sum_multi) -- the regrouped sum over (m,n) of `L(m,n)*w` in
perturbation.jl:394-415"
function spmv_multi(fam::DeviceFamily, c::Vector{ComplexF64}, X::Matrix{ComplexF64})
    size(X, 2) == length(fam.L.terms) == length(c) || error("spmv_multi: one column and one coefficient per term")
    y = Vector{ComplexF64}(undef, size(X, 1))
    check(ccall((:wae_spmv_sum_multi, libwaehip), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}), fam.handle, c, X, y))
    return y
end

# probe_columns / snapshots: workspace hints (opts[8], opts[9] of wae_solver_setup) for the contour integrals that will follow --
# the snapshot store is then mapped during the set-up instead of during the first `compute_moment_matrices`.
function ensure_solver!(fam::DeviceFamily; zref=nothing, opts=Float64[], probe_columns::Int=0, snapshots::Int=0)
    fam.solver_ready && return
    if probe_columns > 0 || snapshots > 0
        opts = vcat(Float64.(opts), fill(0.0, max(0, 8 - length(opts))))[1:8]      # 0: keep the library's default
        opts = vcat(opts, Float64[probe_columns, snapshots])
    end
    L = fam.L
    z = zref === nothing ? L.params[L.eigval] : zref
    isfinite(z) || (z = 0.0im)
    saved = (copy(L.params), L.active, L.mode); L.active = [L.eigval]; L.mode = :all
    c = coefficients(L, z); L.params, L.active, L.mode = saved
    check(ccall((:wae_solver_setup, libwaehip), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{Float64}, Int32), fam.handle, c, opts, length(opts)))
    fam.solver_ready = true
end

function Base.:\(A::Operator, B::StridedVecOrMat{ComplexF64})
    ensure_solver!(A.fam)
    X = similar(B); info = Ref{SolveInfo}()
    check(ccall((:wae_solve, libwaehip), Cint,
                (Ptr{Cvoid}, Ptr{ComplexF64}, Int32, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32, Int32, Float64, Int32, Ref{SolveInfo}),
                A.fam.handle, A.c, 1, B, X, size(B, 2), A.op, A.fam.tol, A.fam.maxit, info))
    report(0, info[], "\\")
    return X
end

"`A\\b` with a known dominant direction of the solution (`u = L(z)\\(L(z,1)*x0)`, iterative_solvers.jl:307,571-572): wae_solve_guess"
function solve_guess(A::Operator, B::StridedVecOrMat{ComplexF64}, G::StridedVecOrMat{ComplexF64}; quiet::Bool=true)
    ensure_solver!(A.fam)
    X = similar(B); info = Ref{SolveInfo}()
    code = check(ccall((:wae_solve_guess, libwaehip), Cint,
                (Ptr{Cvoid}, Ptr{ComplexF64}, Int32, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32, Int32, Float64, Int32, Ref{SolveInfo}),
                A.fam.handle, A.c, 1, B, G, X, size(B, 2), A.op, A.fam.tol, A.fam.maxit, info))
    report(code, info[], "solve_guess"; quiet=quiet)
    return X
end

"res[j] = ||L(ω_j) v_j|| / Σ_k |c_jk| ||A_k v_j||: which Ritz pairs of `beyn` are eigenpairs (wae_eig_residuals)"
function eig_residuals(fam::DeviceFamily, Ω::AbstractVector, P::Matrix{ComplexF64})
    L = fam.L
    saved = (copy(L.params), L.active, L.mode); L.active = [L.eigval]; L.mode = :all
    ct = Matrix{ComplexF64}(undef, length(L.terms), length(Ω))
    for (j, w) in enumerate(Ω); ct[:, j] = coefficients(L, w); end
    L.params, L.active, L.mode = saved
    res = zeros(Float64, length(Ω))
    check(ccall((:wae_eig_residuals, libwaehip), Cint, (Ptr{Cvoid}, Int32, Ptr{ComplexF64}, Ptr{ComplexF64}, UInt64, Ptr{Float64}),
                fam.handle, length(Ω), ct, P, 0, res))
    return res
end

# snapshot points for the projected initial guesses: rb indices spread evenly through the quadrature list, re-ordered so
# that every prefix covers the contour (bit reversal) -- the device takes the snapshots progressively
function _snapshot_split(n::Int, rb::Int)
    rb = min(rb, n)
    idx = unique(floor.(Int, ((0:rb-1) .+ 0.5) .* n ./ rb) .+ 1)
    bits = max(1, ceil(Int, log2(length(idx))))
    key = [parse(Int, reverse(string(i, base=2, pad=bits)), base=2) for i in 0:length(idx)-1]
    return idx[sortperm(key)], setdiff(1:n, idx)
end

"moments of beyn.jl:62-74 / compute_moment_matrices (beyn.jl:251-268) on the device; `rb` = number of snapshot points
for projected initial guesses (wae_beyn_moments_rb; default 40 for contours of at least 64 points)"
function compute_moment_matrices(fam::DeviceFamily, Γ, V::Matrix{ComplexF64}; K=1, N=16, rb=nothing)
    ensure_solver!(fam)
    L = fam.L
    X, W = FastGaussQuadrature.gausslegendre(N)
    zs = ComplexF64[]; ws = ComplexF64[]
    for i in 1:length(Γ)
        a, b = Γ[i], Γ[i == length(Γ) ? 1 : i + 1]
        append!(zs, X .* (b - a) / 2 .+ (a + b) / 2); append!(ws, W .* (b - a) / 2)
    end
    T = length(L.terms)
    saved = (L.active, L.mode); L.active = [L.eigval]; L.mode = :all
    ct = Matrix{ComplexF64}(undef, T, length(zs))                 # column j = coefficients of L(z_j)  (row-major npts x T for C)
    for (j, z) in enumerate(zs); ct[:, j] = coefficients(L, z); end
    L.active, L.mode = saved
    d, l = size(V)
    A = zeros(ComplexF64, d, l, 2K); info = Ref{SolveInfo}()
    npts = length(zs)
    rb === nothing && (rb = (npts >= 64 && d >= 1000) ? min(40, div(npts, 2)) : 0)
    if rb == 0 || npts < 2rb
        check(ccall((:wae_beyn_moments, libwaehip), Cint,
                    (Ptr{Cvoid}, Int32, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32, Int32, Float64, Int32,
                     Ptr{ComplexF64}, UInt64, Ref{SolveInfo}),
                    fam.handle, npts, zs, ws, ct, V, l, K, fam.tol, fam.maxit, A, 0, info))
        report(0, info[], "beyn moments"; fatal=true)
        return A
    end
    idx, rest = _snapshot_split(npts, rb)
    A1 = zeros(ComplexF64, d, l, 2K)
    sig = (Ptr{Cvoid}, Int32, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32, Int32, Float64, Int32,
           Int32, Int32, Int32, UInt64, Ptr{ComplexF64}, UInt64, Int32, Int32, Int32, Ref{SolveInfo})
    # mode 0: the snapshot points (solutions kept in the handle's store); mode 2: all other points from the projection
    check(ccall((:wae_beyn_moments_rb, libwaehip), Cint, sig, fam.handle, length(idx), zs[idx], ws[idx], ct[:, idx], V, l, K,
                fam.tol, fam.maxit, 0, length(idx), 0, 0, A, 0, 0, 0, 0, info))
    report(0, info[], "beyn moments (snapshot points)"; fatal=true)
    # V = C_NULL: the probe matrix uploaded by the mode-0 call is still on the device (include/waehip.h)
    check(ccall((:wae_beyn_moments_rb, libwaehip), Cint, sig, fam.handle, length(rest), zs[rest], ws[rest], ct[:, rest], C_NULL, l, K,
                fam.tol, fam.maxit, 2, length(idx), 0, 0, A1, 0, 0, 0, 0, info))
    report(0, info[], "beyn moments (projected points)"; fatal=true)
    return A .+ A1                                                  # the moments are a plain sum over quadrature points
end

"Ω, P = beyn(Ld, Γ; l, K, N, tol, pos_test) -- src/NLEVP/beyn.jl:34-110 with the quadrature loop on the GPU"
function beyn(fam::DeviceFamily, Γ; l=5, K=1, N=16, tol=0.0, pos_test=true)
    d = size(fam.L.terms[1].coeff, 1)
    K = max(K, div(l, d) + Int(mod(l, d) != 0))
    V = zeros(ComplexF64, d, l); for i in 1:min(d, l); V[i, i] = 1; end
    A = compute_moment_matrices(fam, Γ, V; K=K, N=N)
    B = Array{ComplexF64}(undef, d * K, l * K, 2)
    for i in 0:K-1, j in 0:K-1
        B[(1:d).+d*i, (1:l).+l*j, 1] = A[:, :, i+j+1]; B[(1:d).+d*i, (1:l).+l*j, 2] = A[:, :, i+j+2]
    end
    U, Σ, W = svd(B[:, :, 1])
    if tol > 0; m = Σ .> tol; U, Σ, W = U[:, m], Σ[m], W[:, m]; end
    Ω, P = eigen(U' * B[:, :, 2] * W * Diagonal(1 ./ Σ)); P = U[1:d, :] * P
    if pos_test; m = map(z -> inpoly(z, Γ), Ω); Ω, P = Ω[m], P[:, m]; end
    return Ω, P
end

"H, V of m Arnoldi steps on op(A)^{-1} op(M): the device half of Arpack.eigs(A,M,sigma=0) (Householder.jl:100-101)"
function arnoldi_shiftinvert(A::Operator, M::Operator, m::Integer, v0::Vector{ComplexF64})
    ensure_solver!(A.fam)
    d = length(v0); H = zeros(ComplexF64, m + 1, m); V = zeros(ComplexF64, d, m + 1); info = Ref{SolveInfo}()
    check(ccall((:wae_arnoldi_shiftinvert, libwaehip), Cint,
                (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32, Ptr{ComplexF64}, Int32, Float64, Int32, Ptr{ComplexF64}, Ptr{ComplexF64}, Ref{SolveInfo}),
                A.fam.handle, A.c, M.c, m, v0, A.op, A.fam.tol, A.fam.maxit, H, V, info))
    return H, V
end


# ---------------------------------------------------------------------------------------------------------------
# The solver surface of NLEVP_exports.jl:1-17 for a DeviceFamily.  The reference's householder / mslp / perturb_fast! cannot
# run unchanged on the operator view: they hand the matrix itself to Arpack.eigs and lu (Householder.jl:100-101,115;
# iterative_solvers.jl:132-133; perturbation.jl:385).  These methods keep the reference's signatures, flag conventions and
# parameter-mutation semantics and put every factorisation / solve / SpMV on the device.
# ---------------------------------------------------------------------------------------------------------------
import ..NLEVP: Solution, pade, poly_roots, householder_update
import ..NLEVP: itsol_converged, itsol_maxiter, itsol_slow_convergence, itsol_impossible, itsol_singular_exception,
                itsol_arpack_exception, itsol_isnan, itsol_unknown

struct EigsError <: Exception
    msg::String
end

"`nsys` shift-invert Arnoldi processes in lock-step (wae_arnoldi_shiftinvert_batch): coefficient rows cA, cM (T x nsys),
start vectors V0 (d x nsys).  Returns H (m+1, m, nsys), V (d, m+1, nsys) and the solve statistics."
function arnoldi_batch(fam::DeviceFamily, cA::Matrix{ComplexF64}, cM::Matrix{ComplexF64}, m::Integer, V0::Matrix{ComplexF64}, op::Int32;
                       ritz_tol::Float64=0.0)
    ensure_solver!(fam)
    d, nsys = size(V0)
    H = zeros(ComplexF64, m + 1, m, nsys); V = zeros(ComplexF64, d, m + 1, nsys); info = Ref{SolveInfo}()
    check(ccall((:wae_arnoldi_shiftinvert_batch, libwaehip), Cint,
                (Ptr{Cvoid}, Int32, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32, Ptr{ComplexF64}, Int32, Float64, Int32, Float64,
                 Ptr{ComplexF64}, Ptr{ComplexF64}, Ref{SolveInfo}),
                fam.handle, nsys, cA, cM, m, V0, op, fam.tol, fam.maxit, ritz_tol, H, V, info))
    return H, V, info[]
end

"lam, v[, gap] of `Arpack.eigs(A, M, nev=nev, sigma=0, v0=v0)` (Householder.jl:100-101; iterative_solvers.jl:132-133): short
Arnoldi factorisations of (A - σM)^{-1} M on the device, restarted with the wanted Ritz vectors; Ritz extraction on the host.
σ: a shift far below the spectral gap once |λ| itself has fallen below it (an iterative inner solver needs a regular operator
where the reference relies on UMFPACK factorising the numerically singular A, Householder.jl:145)."
function eigs(A::Operator, M::Operator; nev::Int=1, v0=nothing, tol::Float64=1e-12, maxiter::Int=300, sigma=0.0)
    fam = A.fam; d = size(A)[1]
    step = min(d, max(20, 2nev + 1), max(6, 2nev + 2))
    v = v0 === nothing ? ones(ComplexF64, d) : Vector{ComplexF64}(v0)
    cA = A.c .- sigma .* M.c
    sig_out = A.op == OP_C ? conj(sigma) : sigma
    last = nothing; gap = Inf; total = 0; failed = 0
    while total < maxiter
        H3, V3, info = arnoldi_batch(fam, reshape(cA, :, 1), reshape(M.c, :, 1), step, reshape(v, :, 1), A.op; ritz_tol=(nev == 1 ? tol : 0.0))
        H, V = H3[:, :, 1], V3[:, :, 1]
        total += step
        if info.n_unconverged > 0 && info.relres_max > 1e-4        # inner solves that do not even reach 1e-4: give up like ARPACK
            failed += 1
            failed >= 2 && throw(EigsError("inner solves stalled at relative residual $(info.relres_max)"))
        end
        m = step
        while m > 1 && all(H[:, m] .== 0); m -= 1; end              # steps not taken (early exit on the device)
        taken = m
        for j in 1:m
            if H[j+1, j] == 0; m = j; break; end                    # invariant subspace
        end
        F = eigen(H[1:m, 1:m])
        ord = sortperm(abs.(F.values); rev=true)
        theta, Y = F.values[ord], F.vectors[:, ord]
        k = min(nev, m)
        res = abs(H[m+1, m]) .* abs.(Y[m, 1:k])
        X = V[:, 1:m] * Y[:, 1:k]
        for j in 1:k; X[:, j] ./= norm(X[:, j]); end
        last = (sig_out .+ 1.0 ./ theta[1:k], X)
        m > k && (gap = abs(1.0 / theta[k+1]))
        (all(res .<= tol .* abs.(theta[1:k])) || m < taken || m >= d) && return last[1], last[2], gap
        v = X * ones(ComplexF64, k)
    end
    last === nothing && throw(EigsError("no Ritz pair"))
    return last[1], last[2], gap
end

"λ_k, v_k of `perturb` / `perturb_disk` / `perturb_norm` (perturbation.jl:319-367,374-444,487-560) as ONE device call (wae_perturb).
norm_mode 0/1/2 as in include/waehip.h (+16: eigenvalue series only)."
function perturb_device(fam::DeviceFamily, N::Int, v0::Vector{ComplexF64}, v0Adj::Vector{ComplexF64}, norm_mode::Int; cY=nothing, quiet::Bool=false)
    ensure_solver!(fam)
    L = fam.L; T = length(L.terms); d = length(v0)
    table = zeros(ComplexF64, T, N + 1, N + 1)                     # [(m*(N+1)+n)*T + k] in C order = [k, n+1, m+1] here
    for m in 0:N, n in 0:N-m
        table[:, n+1, m+1] = coefficients(L, m, n)
    end
    lam = zeros(ComplexF64, N + 1); V = zeros(ComplexF64, d, N + 1); info = Ref{SolveInfo}()
    code = check(ccall((:wae_perturb, libwaehip), Cint,
                (Ptr{Cvoid}, Ptr{ComplexF64}, Int32, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32, Ptr{ComplexF64}, Float64, Int32,
                 Ptr{ComplexF64}, Ptr{ComplexF64}, Ref{SolveInfo}),
                fam.handle, table, N, v0, v0Adj, norm_mode, cY === nothing ? C_NULL : cY, fam.tol, fam.maxit, lam, V, info))
    report(code, info[], "perturb (order $N)"; quiet=quiet)
    return lam, [V[:, k] for k in 1:N+1]
end

function _perturb_wrapper!(sol::Solution, fam::DeviceFamily, param::Symbol, N::Int, mode::Symbol, norm_mode::Int; cY=nothing)
    L = fam.L                                                       # LinOpFam.jl:546-560
    active, params, current_mode = L.active, L.params, L.mode
    L.params = sol.params; L.active = [sol.eigval, param]; L.mode = mode
    key = Symbol("$(string(param))/Taylor")
    try
        lam, v = perturb_device(fam, N, Vector{ComplexF64}(sol.v), Vector{ComplexF64}(sol.v_adj), mode == :householder ? 16 : norm_mode;
                                cY=cY, quiet=(mode == :householder))
        lam[1] = sol.params[sol.eigval]
        sol.eigval_pert[key], sol.v_pert[key] = lam, v
    finally
        L.active, L.mode, L.params = active, current_mode, params
    end
    return
end
"perturb!(sol, Ld, param, N; mode)   (LinOpFam.jl:546-560)"
perturb!(sol::Solution, fam::DeviceFamily, param::Symbol, N::Int; mode=:compact) = _perturb_wrapper!(sol, fam, param, N, mode, 0)
"perturb_fast!(sol, Ld, param, N; mode)   (LinOpFam.jl:575-589): needs no multi-index files (deps/build.jl)"
perturb_fast!(sol::Solution, fam::DeviceFamily, param::Symbol, N::Int; mode=:compact) = _perturb_wrapper!(sol, fam, param, N, mode, 1)
"perturb_norm!(sol, Ld, param, N; mode)   (LinOpFam.jl:604-618): Y = -L.terms[end].coeff"
function perturb_norm!(sol::Solution, fam::DeviceFamily, param::Symbol, N::Int; mode=:compact)
    cY = zeros(ComplexF64, length(fam.L.terms)); cY[end] = -1
    _perturb_wrapper!(sol, fam, param, N, mode, 2; cY=cY)
end

# one pass of the loop body shared by householder and mslp (Householder.jl:96-120)
function _aux_step(fam::DeviceFamily, z, order, nev, v0, v0_adj, update, state)
    L = fam.L
    L.params[L.eigval] = z; L.params[L.auxval] = 0
    L.active = [L.eigval]; L.mode = :all
    A = fam(z)
    cM = zeros(ComplexF64, length(L.terms)); cM[end] = -1           # M = -L.terms[end].coeff  (Householder.jl:92)
    M = Operator(fam, cM, OP_N)
    gap_prev, lam_prev = get(state, :gap, Inf), get(state, :lam, Inf)
    sigma = (isfinite(gap_prev) && lam_prev < 1e-4 * gap_prev) ? 1e-5 * gap_prev : 0.0
    lam, v, gap = eigs(A, M; nev=nev, v0=v0, sigma=sigma)
    lam_adj, v_adj, _ = eigs(A', M'; nev=nev, v0=v0_adj, sigma=sigma)
    state[:gap] = isfinite(gap) ? gap : gap_prev
    state[:lam] = minimum(abs.(lam))
    p = sortperm(abs.(lam)); lam, v = lam[p], v[:, p]
    p = sortperm(abs.(lam_adj)); v_adj = v_adj[:, p]
    cand = ComplexF64[]
    for i in 1:min(nev, length(lam), size(v_adj, 2))
        L.params[L.auxval] = lam[i]
        sol = Solution(L.params, v[:, i], v_adj[:, i], L.auxval)
        perturb!(sol, fam, L.eigval, order; mode=:householder)
        push!(cand, update(sol.eigval_pert[Symbol("$(string(L.eigval))/Taylor")]))
    end
    return lam, v, v_adj, cand
end

function _normalise(fam::DeviceFamily, v0, v0_adj)                 # Householder.jl:189-190
    L = fam.L
    cM = zeros(ComplexF64, length(L.terms)); cM[end] = -1
    v0 = v0 ./ sqrt(dot(v0, Operator(fam, cM, OP_N) * v0))
    saved = (L.active, L.mode); L.active = [L.eigval]; L.mode = :all
    v0_adj = v0_adj ./ conj(dot(v0_adj, fam(L.params[L.eigval], 1) * v0))
    L.active, L.mode = saved
    return v0, v0_adj
end

"sol, n, flag = householder(Ld, z; maxiter, tol, relax, lam_tol, order, nev, v0, v0_adj, output)   (Householder.jl:70-192).
nev = 1 (the default) runs the device-resident iteration of `householder_many` for the one start value (the eigenvector pair stays in HBM
between the Arnoldi processes, the perturbation step and the update: half the time per call at 1M DoF); resident = false or nev > 1:
the vectors pass through host memory between the library calls."
function householder(fam::DeviceFamily, z; maxiter=10, tol=0., relax=1., lam_tol=Inf, order=1, nev=1, v0=[], v0_adj=[], output=false, resident::Bool=true)
    if resident && nev == 1
        d1 = size(fam.L.terms[1].coeff, 1)
        return householder_many(fam, [z]; maxiter=maxiter, tol=tol, relax=relax, lam_tol=lam_tol, order=order,
                                v0s=(v0 == [] ? nothing : reshape(Vector{ComplexF64}(v0), d1, 1)),
                                v0s_adj=(v0_adj == [] ? nothing : reshape(Vector{ComplexF64}(v0_adj), d1, 1)), output=output)[1]
    end
    L = fam.L
    z = ComplexF64(z); z0 = complex(Inf); lam = Inf; n = 0
    active, mode = L.active, L.mode
    d = size(L.terms[1].coeff, 1)
    v0 == [] && (v0 = ones(ComplexF64, d))
    v0_adj == [] && (v0_adj = conj.(v0))
    flag = 1; state = Dict{Symbol,Float64}()
    try
        while abs(z - z0) > tol && n < maxiter
            output && println(n, "\t\t", abs(lam), "\t", abs(z - z0), "\t", z)
            z0 = z
            lams, v, v_adj, dzs = _aux_step(fam, z, order, nev, v0, v0_adj, c -> householder_update([factorial(i - 1) * c[i] for i in 1:length(c)]), state)
            i = sortperm(abs.(dzs))[1]
            lam = lams[i]; L.params[L.auxval] = lam
            z += relax * dzs[i]
            v0 = (1 - relax) .* v0 .+ relax .* v[:, i]
            v0_adj = (1 - relax) .* v0_adj .+ relax .* v_adj[:, i]
            n += 1
        end
    catch excp
        if excp isa EigsError; flag = -4
        elseif excp isa LinearAlgebra.SingularException; flag = -6; L.params[L.eigval] = z
        else; flag = -2; L.params[L.eigval] = z; end
    end
    if flag == 1
        L.params[L.eigval] = z
        flag = n >= maxiter ? -1 : (abs(lam) <= lam_tol ? 1 : (abs(z - z0) <= tol ? 0 : (isnan(z) ? -5 : -3)))
    end
    L.active, L.mode = active, mode
    v0, v0_adj = _normalise(fam, v0, v0_adj)
    return Solution(L.params, v0, v0_adj, L.eigval), n, flag
end

# the single-start iterations on device-resident vectors (slots 4-7, column 1): the pair (v0, v0_adj) of `mslp` stays in HBM
const _SV, _SW, _SXR, _SXL = 4, 5, 6, 7
function _slots_begin(fam::DeviceFamily, v0, v0_adj)
    d = size(fam.L.terms[1].coeff, 1)
    slot_write(fam, _SV, v0 == [] ? ones(ComplexF64, d, 1) : reshape(Vector{ComplexF64}(v0), d, 1))
    if v0_adj == []
        slot_write(fam, _SW, nothing; ncols_total=1)
        slot_axpby(fam, _SW, [1], _SV, [1], 1.0, 0.0; conj_src=true)          # conj(v0)  (Householder.jl:84-86)
    else
        slot_write(fam, _SW, reshape(Vector{ComplexF64}(v0_adj), d, 1))
    end
    slot_write(fam, _SXR, nothing; ncols_total=1); slot_write(fam, _SXL, nothing; ncols_total=1)
    return
end
"`_aux_step` for nev = 1 on the slots: Ritz vectors to _SXR / _SXL, the perturbation step reads them there; returns (lam, candidate update)"
function _aux_step_slots(fam::DeviceFamily, z, order, update, state)
    L = fam.L; T = length(L.terms)
    L.params[L.eigval] = z; L.params[L.auxval] = 0
    L.active = [L.eigval]; L.mode = :all
    cA = reshape(coefficients(L, z), T, 1)
    cM = zeros(ComplexF64, T); cM[end] = -1                         # M = -L.terms[end].coeff  (Householder.jl:92)
    gap_prev, lam_prev = get(state, :gap, Inf), get(state, :lam, Inf)
    sigma = (isfinite(gap_prev) && lam_prev < 1e-4 * gap_prev) ? 1e-5 * gap_prev : 0.0
    right = eigs_many_slots(fam, cA, cM, _SV, [1], OP_N, [sigma], _SXR)[1]
    left = eigs_many_slots(fam, cA, cM, _SW, [1], OP_C, [sigma], _SXL)[1]
    right isa EigsError && throw(right)
    left isa EigsError && throw(left)
    lam, gap = right
    state[:gap] = isfinite(gap) ? gap : gap_prev
    state[:lam] = abs(lam)
    L.params[L.auxval] = lam
    return lam, update(eigval_series_slots(fam, L.auxval, L.eigval, order, _SXR, 1, _SXL, 1))
end
"`_normalise` (Householder.jl:189-190) on the slots, then the pair back to the host"
function _slots_finish(fam::DeviceFamily)
    L = fam.L; T = length(L.terms)
    cM = zeros(ComplexF64, T); cM[end] = -1
    nv = slot_forms(fam, reshape(cM, T, 1), _SV, [1], _SV, [1])
    slot_axpby(fam, _SV, [1], _SV, [1], 1.0 / sqrt(nv[1]), 0.0)
    saved = (L.active, L.mode); L.active = [L.eigval]; L.mode = :all
    cD = reshape(coefficients(L, L.params[L.eigval], 1), T, 1)
    L.active, L.mode = saved
    dw = slot_forms(fam, cD, _SW, [1], _SV, [1])
    slot_axpby(fam, _SW, [1], _SW, [1], 1.0 / conj(dw[1]), 0.0)
    return slot_read(fam, _SV, 1, 1)[:, 1], slot_read(fam, _SW, 1, 1)[:, 1]
end

"sol, n, flag = mslp(Ld, z; maxiter, tol, relax, lam_tol, order, nev, v0, v0_adj, num_order, scale, output)   (iterative_solvers.jl:93-252).
The family must end with its auxiliary term (`discretize` adds it, Helmholtz.jl:571-574): the device copy is immutable."
function mslp(fam::DeviceFamily, z; maxiter=10, tol=0., relax=1., lam_tol=Inf, order=1, nev=1, v0=[], v0_adj=[], num_order=1, scale=1, output=false,
              resident::Bool=true)
    L = fam.L
    L.terms[end].operator == "__aux__" || error("mslp(::DeviceFamily): push the __aux__ term before creating the DeviceFamily (iterative_solvers.jl:119-123)")
    z = ComplexF64(z) * scale; tol *= scale
    z0 = complex(Inf); lam = Inf; lam0 = complex(Inf); n = 0
    active, mode = L.active, L.mode
    d = size(L.terms[1].coeff, 1)
    on_dev = resident && nev == 1                                   # the pair stays in HBM (slots) between the library calls
    if on_dev
        ensure_solver!(fam); _slots_begin(fam, v0, v0_adj)
    else
        v0 == [] && (v0 = ones(ComplexF64, d))
        v0_adj == [] && (v0_adj = conj.(v0))
    end
    flag = itsol_converged; state = Dict{Symbol,Float64}()
    polyval(p, x) = foldr((a, acc) -> a + x * acc, p)
    try
        while abs(z - z0) > tol && n < maxiter
            output && println(n, "\t\t", abs(z - z0) / scale, "\t", z / scale)
            pades = Tuple{Vector{ComplexF64},Vector{ComplexF64}}[]
            upd = function (c)
                num, den = pade(c, num_order, order - num_order)
                push!(pades, (num, den))
                r = poly_roots(num)
                return r[sortperm(abs.(r))[1]]
            end
            local lams, v, v_adj, dzs
            if on_dev
                lam1, dz1 = _aux_step_slots(fam, z, order, upd, state)
                lams, dzs = [lam1], [dz1]
            else
                lams, v, v_adj, dzs = _aux_step(fam, z, order, nev, v0, v0_adj, upd, state)
            end
            i = isinf(z0) ? sortperm(abs.(dzs))[1] : sortperm([abs(lam0 - polyval(nd[1], z0 - z) / polyval(nd[2], z0 - z)) for nd in pades])[1]
            lam = lams[i]; L.params[L.auxval] = lam
            z0 = z; lam0 = lam
            z += relax * dzs[i]
            if on_dev
                slot_axpby(fam, _SV, [1], _SXR, [1], relax, 1 - relax); slot_axpby(fam, _SW, [1], _SXL, [1], relax, 1 - relax)
            else
                v0 = (1 - relax) .* v0 .+ relax .* v[:, i]
                v0_adj = (1 - relax) .* v0_adj .+ relax .* v_adj[:, i]
            end
            n += 1
        end
    catch excp
        if excp isa EigsError; flag = itsol_arpack_exception
        elseif excp isa LinearAlgebra.SingularException; flag = itsol_singular_exception; L.params[L.eigval] = z
        else; flag = itsol_unknown; L.params[L.eigval] = z; end
    end
    if flag == itsol_converged
        L.params[L.eigval] = z
        flag = n >= maxiter ? itsol_maxiter : (abs(lam) <= lam_tol ? itsol_converged : (abs(z - z0) <= tol ? itsol_slow_convergence : (isnan(z) ? itsol_isnan : itsol_impossible)))
    end
    L.active, L.mode = active, mode
    v0, v0_adj = on_dev ? _slots_finish(fam) : _normalise(fam, v0, v0_adj)
    return Solution(L.params, v0, v0_adj, L.eigval), n, flag
end

"sol, n, flag = inveriter(Ld, z; maxiter, tol, relax, x0, v, output)   (iterative_solvers.jl:285-347): u = L(z)\\(L(z,1) x0) on the device,
with the current iterate deflated as the known dominant direction of the solution (wae_solve_guess)"
function inveriter(fam::DeviceFamily, z; maxiter=10, tol=0., relax=1., x0=[], v=[], output=false)
    L = fam.L
    d = size(L.terms[1].coeff, 1)
    x0 == [] && (x0 = ones(ComplexF64, d)); v == [] && (v = ones(ComplexF64, d))
    z = ComplexF64(z); z0 = complex(Inf); n = 0; flag = itsol_converged
    active, mode = L.active, L.mode; L.active = [L.eigval]; L.mode = :all
    try
        while abs(z - z0) > tol && n < maxiter
            output && println(n, "\t\t", abs(z - z0), "\t", z)
            z0 = z
            u = solve_guess(fam(z), fam(z, 1) * x0, x0)
            z = z0 - dot(v, x0) / dot(v, u)
            x0 = u ./ dot(v, u)
            n += 1
        end
    catch excp
        flag = excp isa LinearAlgebra.SingularException ? itsol_singular_exception : itsol_unknown
    end
    if flag == itsol_converged
        flag = n >= maxiter ? itsol_maxiter : (abs(z - z0) <= tol ? itsol_converged : (isnan(z) ? itsol_isnan : itsol_impossible))
    end
    L.params[L.eigval] = z
    L.active, L.mode = active, mode
    return Solution(L.params, x0, ComplexF64[], L.eigval), n, flag
end

"Dict ω => [Solution, inside] = solve(Ld, Γ; Δl, N, tol, eigvals, maxcycles, nev, max_outer_cycles, atol_σ, rtol_σ, loglevel)
(solver.jl:36-184): Beyn with growing probe blocks, analytic deflation of the known eigenpairs from the moments, local refinement
(the reference calls the un-included `mehrmann` at :106; `inveriter` is the same algorithm, mehrmann.jl:1-72)"
function solve(fam::DeviceFamily, Γ; Δl=1, N=16, tol=1e-8, eigvals=Dict(), maxcycles=1, nev=1, max_outer_cycles=1, atol_σ=1e-12, rtol_σ=1e-8, loglevel=0)
    L = fam.L
    d = size(L.terms[1].coeff, 1)
    eigvals = Dict{ComplexF64,Any}(eigvals)
    A = Array{Array{ComplexF64,3},1}()
    l = 0
    for cycle in 1:max_outer_cycles * max(1, div(d, Δl))
        l >= d && break
        V = zeros(ComplexF64, d, Δl); for i in 1:Δl; l + i <= d && (V[l+i, i] = 1); end
        mom = compute_moment_matrices(fam, Γ, V; K=1, N=N)
        for (ω, val) in eigvals                                    # deflation of known eigenpairs (solver.jl:57-64,131-137)
            sol = val[1]; val[2] || continue
            for p in 0:size(mom, 3)-1
                mom[:, :, p+1] .-= (-2π * im * ω^p) .* (sol.v * (sol.v_adj[l+1:l+Δl])')
            end
        end
        push!(A, mom); l += Δl
        B0 = hcat((a[:, :, 1] for a in A)...); B1 = hcat((a[:, :, 2] for a in A)...)
        F = svd(B0)
        σmax = isempty(F.S) ? 0.0 : F.S[1]
        keep = (F.S .> atol_σ) .& (F.S .> rtol_σ * σmax)
        any(keep) || break
        U, S, W = F.U[:, keep], F.S[keep], F.V[:, keep]
        Ω, P = eigen(U' * B1 * W * Diagonal(1 ./ S)); P = U * P
        found_new = false
        for (j, ω0) in enumerate(Ω)
            inpoly(ω0, Γ) || continue
            sol, n, flag = inveriter(fam, ω0; maxiter=10, tol=tol, x0=P[:, j], v=P[:, j])
            flag == itsol_converged || continue
            ω = sol.params[L.eigval]
            any(abs(ω - w) <= 10tol * max(1, abs(w)) for w in keys(eigvals)) && continue
            # adjoint vector by one more inverse iteration on L(ω)' (normalised like Householder.jl:189-190)
            y = solve_guess(fam(ω)', (fam(ω, 1))' * conj.(sol.v), conj.(sol.v))
            sol.v_adj = y ./ conj(dot(y, fam(ω, 1) * sol.v))
            eigvals[ω] = Any[sol, inpoly(ω, Γ)]
            found_new = true
            loglevel > 0 && println("solve: new eigenvalue ", ω, " after ", n, " iterations")
        end
        (!found_new && sum(keep) < l) && break                     # rank gap reached and nothing new: done (solver.jl:172)
    end
    return eigvals
end

# ---------------------------------------------------------------------------------------------------------------
# householder for SEVERAL start values in lock-step (the refinement `solve` wants after `beyn`, solver.jl:96-140): the two
# shift-invert Arnoldi processes of every Newton step are batched over the start values on the device.  A single-column solve is
# latency-bound: refining the 8 estimates of the 1M-DoF benchmark this way costs about what refining one does.
# ---------------------------------------------------------------------------------------------------------------
"per-system (lam, X, gap) of `eigs` for nsys operator pairs (A_s - σ_s M, M) in lock-step: cA T x nsys, v0 d x nsys.
An entry is an EigsError when that system's inner solves stalled."
function eigs_many(fam::DeviceFamily, cA::Matrix{ComplexF64}, cM::Vector{ComplexF64}, V0::Matrix{ComplexF64}, op::Int32, sigmas::Vector{Float64};
                   nev::Int=1, tol::Float64=1e-12, maxiter::Int=300)
    d, nsys = size(V0)
    step = min(d, max(20, 2nev + 1), max(6, 2nev + 2))
    cAs = cA .- reshape(ComplexF64.(sigmas), 1, :) .* cM
    sig_out = op == OP_C ? conj.(ComplexF64.(sigmas)) : ComplexF64.(sigmas)
    V0 = copy(V0)
    out = Vector{Any}(undef, nsys)
    pending = collect(1:nsys); total = 0
    while !isempty(pending) && total < maxiter
        H, V, info = arnoldi_batch(fam, cAs[:, pending], repeat(cM, 1, length(pending)), step, V0[:, pending], op; ritz_tol=(nev == 1 ? tol : 0.0))
        total += step
        failed = info.n_unconverged > 0 && info.relres_max > 1e-4
        still = Int[]
        for (q, s) in enumerate(pending)
            Hs = H[:, :, q]; Vs = V[:, :, q]
            m = step
            while m > 1 && all(Hs[:, m] .== 0); m -= 1; end            # steps not taken (early exit on the device)
            taken = m
            for j in 1:m
                if Hs[j+1, j] == 0; m = j; break; end                  # invariant subspace
            end
            F = eigen(Hs[1:m, 1:m])
            ord = sortperm(abs.(F.values); rev=true)
            theta, Y = F.values[ord], F.vectors[:, ord]
            k = min(nev, m)
            res = abs(Hs[m+1, m]) .* abs.(Y[m, 1:k])
            X = Vs[:, 1:m] * Y[:, 1:k]
            for j in 1:k; X[:, j] ./= norm(X[:, j]); end
            gap = m > k ? abs(1.0 / theta[k+1]) : Inf
            out[s] = (sig_out[s] .+ 1.0 ./ theta[1:k], X, gap)
            if !(all(res .<= tol .* abs.(theta[1:k])) || m < taken || m >= d)
                if failed
                    out[s] = EigsError("inner solves stalled")
                else
                    V0[:, s] = X * ones(ComplexF64, k); push!(still, s)
                end
            end
        end
        pending = still
    end
    return out
end

"start vectors of the left (adjoint) processes when the caller gives none: conj(v) for an isolated mode (Householder.jl:84-86); for
several start vectors their conjugate span, bi-orthogonal in the bilinear form, W = conj(V G^-1), G = transpose(V) V -- a spinning
mode of an annulus has vᵀv = 0 and conj(v) is its PARTNER, orthogonal to the left vector wanted."
function conjugate_span_start(V::Matrix{ComplexF64})
    ns = size(V, 2)
    ns < 2 && return conj.(V)
    nrm = [norm(V[:, j]) for j in 1:ns]; nrm[nrm .== 0] .= 1.0
    Vn = V ./ reshape(nrm, 1, :)
    G = transpose(Vn) * Vn
    (all(isfinite, G) && minimum(svdvals(G)) >= 1e-6) || return conj.(V)
    return conj.(Vn * inv(G))
end

"`householder_many` with every vector passing through host memory between the device calls (wae_arnoldi_shiftinvert_batch,
wae_perturb, wae_spmv_sum_cols): the cross-check of the device-resident form below, `householder_many(...; resident=false)`."
function householder_many_host(fam::DeviceFamily, zs; maxiter=10, tol=0., relax=1., lam_tol=Inf, order=1, v0s=nothing, v0s_adj=nothing, output=false)
    L = fam.L
    z = ComplexF64.(collect(zs)); ns = length(z)
    ns == 0 && return Tuple{Solution,Int,Int}[]
    ensure_solver!(fam)
    d = size(L.terms[1].coeff, 1); T = length(L.terms)
    active, mode = L.active, L.mode
    V = v0s === nothing ? ones(ComplexF64, d, ns) : Matrix{ComplexF64}(reshape(v0s, d, ns))
    W = v0s_adj === nothing ? conjugate_span_start(V) : Matrix{ComplexF64}(reshape(v0s_adj, d, ns))
    z0 = fill(complex(Inf), ns); lam = fill(complex(Inf), ns); n = zeros(Int, ns); flag = ones(Int, ns)
    gaps = fill(Inf, ns); lams = fill(Inf, ns)
    cM = zeros(ComplexF64, T); cM[end] = -1                             # M = -L.terms[end].coeff  (Householder.jl:92)
    upd(c) = householder_update([factorial(i - 1) * c[i] for i in 1:length(c)])
    while true
        act = [s for s in 1:ns if flag[s] == 1 && abs(z[s] - z0[s]) > tol && n[s] < maxiter]
        isempty(act) && break
        cA = Matrix{ComplexF64}(undef, T, length(act)); sig = Float64[]
        L.active = [L.eigval]; L.mode = :all
        for (q, s) in enumerate(act)
            z0[s] = z[s]
            L.params[L.eigval] = z[s]; L.params[L.auxval] = 0
            cA[:, q] = coefficients(L, z[s])
            push!(sig, (isfinite(gaps[s]) && lams[s] < 1e-4 * gaps[s]) ? 1e-5 * gaps[s] : 0.0)
        end
        local right, left
        try
            right = eigs_many(fam, cA, cM, V[:, act], OP_N, sig)
            left = eigs_many(fam, cA, cM, W[:, act], OP_C, sig)
        catch excp
            for s in act; flag[s] = excp isa LinearAlgebra.SingularException ? -6 : -2; end
            break
        end
        for (q, s) in enumerate(act)
            if right[q] isa EigsError || left[q] isa EigsError; flag[s] = -4; continue; end
            lam_r, v_r, gap = right[q]; _, v_l, _ = left[q]
            isfinite(gap) && (gaps[s] = gap)
            lams[s] = minimum(abs.(lam_r))
            L.params[L.eigval] = z[s]; L.params[L.auxval] = lam_r[1]
            local dz
            try
                sol = Solution(L.params, v_r[:, 1], v_l[:, 1], L.auxval)
                perturb!(sol, fam, L.eigval, order; mode=:householder)
                dz = upd(sol.eigval_pert[Symbol("$(string(L.eigval))/Taylor")])
            catch excp
                flag[s] = excp isa LinearAlgebra.SingularException ? -6 : -2
                continue
            end
            lam[s] = lam_r[1]
            output && println(s, " ", n[s], "\t", abs(lam[s]), "\t", abs(dz), "\t", z[s])
            z[s] += relax * dz
            V[:, s] = (1 - relax) .* V[:, s] .+ relax .* v_r[:, 1]
            W[:, s] = (1 - relax) .* W[:, s] .+ relax .* v_l[:, 1]
            n[s] += 1
        end
    end
    # Householder.jl:189-190 for all start values at once: two batched operator products
    MV = Operator(fam, cM, OP_N) * V
    for s in 1:ns; V[:, s] ./= sqrt(dot(V[:, s], MV[:, s])); end
    cD = Matrix{ComplexF64}(undef, T, ns)
    saved = copy(L.params); L.active = [L.eigval]; L.mode = :all
    for s in 1:ns
        L.params[L.eigval] = z[s]; L.params[L.auxval] = isfinite(lam[s]) ? lam[s] : 0
        cD[:, s] = coefficients(L, z[s], 1)
    end
    merge!(L.params, saved); L.active, L.mode = active, mode
    DV = spmv_cols(fam, cD, V)
    for s in 1:ns; W[:, s] ./= conj(dot(W[:, s], DV[:, s])); end
    out = Tuple{Solution,Int,Int}[]
    for s in 1:ns
        f = flag[s]
        L.params[L.eigval] = z[s]; L.params[L.auxval] = isfinite(lam[s]) ? lam[s] : 0
        if f == 1
            f = n[s] >= maxiter ? -1 : (abs(lam[s]) <= lam_tol ? 1 : (abs(z[s] - z0[s]) <= tol ? 0 : (isnan(z[s]) ? -5 : -3)))
        end
        push!(out, (Solution(L.params, V[:, s], W[:, s], L.eigval), n[s], f))
    end
    L.active, L.mode = active, mode
    return out
end

# ---------------------------------------------------------------------------------------------------------------
# Device-resident multivectors ("slots", include/waehip.h): the vectors of the lock-step Newton iteration stay in HBM between the
# calls.  Column indices are 0-based in the library; the wrappers below take Julia's 1-based indices.
# ---------------------------------------------------------------------------------------------------------------
const NSLOTS = 8
_cols0(cols) = Int32[Int32(c - 1) for c in cols]

"X (d x n) into columns col0, col0+1, ... (1-based) of the slot, (re)created with ncols_total columns if its width differs; X = nothing: create / resize only"
function slot_write(fam::DeviceFamily, slot::Integer, X::Union{Nothing,Matrix{ComplexF64}}; ncols_total::Integer=(X === nothing ? 0 : size(X, 2)), col0::Integer=1)
    n = X === nothing ? 0 : size(X, 2)
    check(ccall((:wae_slot_write, libwaehip), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ptr{ComplexF64}),
                fam.handle, slot, ncols_total, col0 - 1, n, X === nothing ? C_NULL : X))
    return
end
function slot_read(fam::DeviceFamily, slot::Integer, col0::Integer, ncols::Integer)
    d = size(fam.L.terms[1].coeff, 1)
    X = Matrix{ComplexF64}(undef, d, ncols)
    check(ccall((:wae_slot_read, libwaehip), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{ComplexF64}), fam.handle, slot, col0 - 1, ncols, X))
    return X
end
"dst[:, dst_cols[i]] = alpha[i] src[:, src_cols[i]] + beta[i] dst[:, dst_cols[i]], one column after the other (conj_src: conj of the source column)"
function slot_axpby(fam::DeviceFamily, dst_slot::Integer, dst_cols, src_slot::Integer, src_cols, alpha, beta; conj_src::Bool=false)
    n = length(dst_cols)
    a = alpha isa Number ? fill(ComplexF64(alpha), n) : Vector{ComplexF64}(alpha)
    b = beta isa Number ? fill(ComplexF64(beta), n) : Vector{ComplexF64}(beta)
    check(ccall((:wae_slot_axpby, libwaehip), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Int32, Ptr{Int32}, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32),
                fam.handle, n, dst_slot, _cols0(dst_cols), src_slot, _cols0(src_cols), a, b, conj_src ? 1 : 0))
    return
end
"out[i] = a_i' op(sum_k C[k, i] A_k) b_i for pairs of slot columns (C: T x n)"
function slot_forms(fam::DeviceFamily, C::Matrix{ComplexF64}, a_slot::Integer, a_cols, b_slot::Integer, b_cols; op::Int32=OP_N)
    n = length(a_cols)
    out = zeros(ComplexF64, n)
    check(ccall((:wae_slot_forms, libwaehip), Cint, (Ptr{Cvoid}, Int32, Ptr{ComplexF64}, Int32, Int32, Ptr{Int32}, Int32, Ptr{Int32}, Ptr{ComplexF64}),
                fam.handle, n, C, op, a_slot, _cols0(a_cols), b_slot, _cols0(b_cols), out))
    return out
end
"`arnoldi_batch` with the start vectors in slot columns and the basis kept on the device: returns H (m+1, m, nsys) and the solve statistics"
function arnoldi_slots(fam::DeviceFamily, cA::Matrix{ComplexF64}, cM::Matrix{ComplexF64}, m::Integer, v0_slot::Integer, v0_cols, op::Int32;
                       ritz_tol::Float64=0.0)
    ensure_solver!(fam)
    nsys = length(v0_cols)
    H = zeros(ComplexF64, m + 1, m, nsys); info = Ref{SolveInfo}()
    check(ccall((:wae_arnoldi_shiftinvert_slots, libwaehip), Cint,
                (Ptr{Cvoid}, Int32, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32, Int32, Ptr{Int32}, Int32, Float64, Int32, Float64,
                 Ptr{ComplexF64}, Ref{SolveInfo}),
                fam.handle, nsys, cA, cM, m, v0_slot, _cols0(v0_cols), op, fam.tol, fam.maxit, ritz_tol, H, info))
    return H, info[]
end
"dst[:, dst_cols[s]] = sum_j Y[j, s] v_j^(s) of the basis of the last arnoldi_slots call (Y: ny x nsys)"
function ritz_to_slot(fam::DeviceFamily, Y::Matrix{ComplexF64}, dst_slot::Integer, dst_cols; normalise::Bool=true)
    ny, nsys = size(Y)
    check(ccall((:wae_arnoldi_ritz_to_slot, libwaehip), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{ComplexF64}, Int32, Ptr{Int32}, Int32),
                fam.handle, nsys, ny, Y, dst_slot, _cols0(dst_cols), normalise ? 1 : 0))
    return
end
"the eigenvalue series of `perturb!(sol, L, param, N; mode = :householder)` (LinOpFam.jl:546-560, perturbation.jl:319-367) for the
eigenpair in slot columns: L.params carries the expansion point, `eigval` names the pair's eigenvalue parameter.  No vector leaves the device."
function eigval_series_slots(fam::DeviceFamily, eigval::Symbol, param::Symbol, N::Int, v_slot::Integer, v_col::Integer, w_slot::Integer, w_col::Integer)
    ensure_solver!(fam)
    L = fam.L; T = length(L.terms)
    active, current_mode = L.active, L.mode
    L.active = [eigval, param]; L.mode = :householder
    table = zeros(ComplexF64, T, N + 1, N + 1)
    try
        for m in 0:N, n in 0:N-m
            table[:, n+1, m+1] = coefficients(L, m, n)
        end
    finally
        L.active, L.mode = active, current_mode
    end
    lam = zeros(ComplexF64, N + 1); info = Ref{SolveInfo}()
    code = check(ccall((:wae_perturb_slots, libwaehip), Cint,
                (Ptr{Cvoid}, Ptr{ComplexF64}, Int32, Int32, Int32, Int32, Int32, Int32, Ptr{ComplexF64}, Float64, Int32,
                 Ptr{ComplexF64}, Ptr{ComplexF64}, Ref{SolveInfo}),
                fam.handle, table, N, v_slot, v_col - 1, w_slot, w_col - 1, 16, C_NULL, fam.tol, fam.maxit, lam, C_NULL, info))
    report(code, info[], "perturb_slots (order $N)"; quiet=true)
    lam[1] = L.params[eigval]
    return lam
end

"`eigs_many` (nev = 1) on device-resident vectors: start vectors = columns `cols` of slot v0_slot, the normalised Ritz vector of system q
goes to column cols[q] of out_slot.  Returns per system (lam, gap) or an EigsError."
function eigs_many_slots(fam::DeviceFamily, cA::Matrix{ComplexF64}, cM::Vector{ComplexF64}, v0_slot::Integer, cols::Vector{Int}, op::Int32,
                         sigmas::Vector{Float64}, out_slot::Integer; tol::Float64=1e-12, maxiter::Int=300)
    nsys = length(cols); d = size(fam.L.terms[1].coeff, 1)
    step = min(d, 6)
    cAs = cA .- reshape(ComplexF64.(sigmas), 1, :) .* cM
    sig_out = op == OP_C ? conj.(ComplexF64.(sigmas)) : ComplexF64.(sigmas)
    out = Vector{Any}(undef, nsys)
    pending = collect(1:nsys); total = 0; src = v0_slot
    while !isempty(pending) && total < maxiter
        H, info = arnoldi_slots(fam, cAs[:, pending], repeat(cM, 1, length(pending)), step, src, cols[pending], op; ritz_tol=tol)
        total += step
        failed = info.n_unconverged > 0 && info.relres_max > 1e-4
        Y = zeros(ComplexF64, step + 1, length(pending)); ny = 1
        still = Int[]
        for (q, s) in enumerate(pending)
            Hs = H[:, :, q]
            m = step
            while m > 1 && all(Hs[:, m] .== 0); m -= 1; end            # steps not taken (early exit on the device)
            taken = m
            for j in 1:m
                if Hs[j+1, j] == 0; m = j; break; end                  # invariant subspace
            end
            F = eigen(Hs[1:m, 1:m])
            ord = sortperm(abs.(F.values); rev=true)
            theta, Yr = F.values[ord], F.vectors[:, ord]
            res = abs(Hs[m+1, m]) * abs(Yr[m, 1])
            Y[1:m, q] = Yr[:, 1]; ny = max(ny, m)
            gap = m > 1 ? abs(1.0 / theta[2]) : Inf
            out[s] = (sig_out[s] + 1.0 / theta[1], gap)
            if !(res <= tol * abs(theta[1]) || m < taken || m >= d)
                if failed
                    out[s] = EigsError("inner solves stalled")
                else
                    push!(still, s)                                     # restart from the Ritz vector (now in out_slot)
                end
            end
        end
        ritz_to_slot(fam, Y[1:ny, :], out_slot, cols[pending])
        pending = still; src = out_slot
    end
    return out
end

"C (ns x ns) with W = conj(V C) the conjugate-span start of the left processes (see conjugate_span_start), or nothing for W = conj(V)"
function conjugate_span_coefficients(V::Matrix{ComplexF64})
    ns = size(V, 2)
    ns < 2 && return nothing
    nrm = [norm(view(V, :, j)) for j in 1:ns]; nrm[nrm .== 0] .= 1.0
    G = (transpose(V) * V) ./ (nrm * transpose(nrm))
    (all(isfinite, G) && minimum(svdvals(G)) >= 1e-6) || return nothing
    return inv(G) ./ nrm
end

"[(sol, n, flag), ...] = householder_many(Ld, zs; maxiter, tol, relax, lam_tol, order, v0s, v0s_adj): `householder`
(Householder.jl:70-192, nev = 1) for every start value in zs, the device work batched over the start values and every vector of the
iteration resident in HBM (slots 5-8 of the family): the estimates go to the device once, the Arnoldi processes start from slot columns
and leave their Ritz vectors there, the perturbation step reads them there, the relaxed update and the normalisations of
Householder.jl:173-176,189-190 are slot operations, and the eigenvectors come back once at the end.  An empty zs returns an empty list.
resident = false: the same iteration through host memory (householder_many_host)."
function householder_many(fam::DeviceFamily, zs; maxiter=10, tol=0., relax=1., lam_tol=Inf, order=1, v0s=nothing, v0s_adj=nothing, output=false,
                          resident::Bool=true)
    resident || return householder_many_host(fam, zs; maxiter=maxiter, tol=tol, relax=relax, lam_tol=lam_tol, order=order, v0s=v0s,
                                             v0s_adj=v0s_adj, output=output)
    L = fam.L
    z = ComplexF64.(collect(zs)); ns = length(z)
    ns == 0 && return Tuple{Solution,Int,Int}[]
    ensure_solver!(fam)
    d = size(L.terms[1].coeff, 1); T = length(L.terms)
    active, mode = L.active, L.mode
    SV, SW, SXR, SXL = 4, 5, 6, 7                                       # (0-based slot numbers: the upper half, 0-3 stay the caller's)
    allc = collect(1:ns)
    V0 = v0s === nothing ? ones(ComplexF64, d, ns) : Matrix{ComplexF64}(reshape(v0s, d, ns))
    slot_write(fam, SV, V0)
    if v0s_adj === nothing
        Cs = v0s === nothing ? nothing : conjugate_span_coefficients(V0)
        slot_write(fam, SW, nothing; ncols_total=ns)
        if Cs === nothing
            slot_axpby(fam, SW, allc, SV, allc, 1.0, 0.0; conj_src=true)
        else
            for i in 1:ns
                slot_axpby(fam, SW, allc, SV, fill(i, ns), conj.(Cs[i, :]), i == 1 ? 0.0 : 1.0; conj_src=true)
            end
        end
    else
        slot_write(fam, SW, Matrix{ComplexF64}(reshape(v0s_adj, d, ns)))
    end
    slot_write(fam, SXR, nothing; ncols_total=ns)
    slot_write(fam, SXL, nothing; ncols_total=ns)
    z0 = fill(complex(Inf), ns); lam = fill(complex(Inf), ns); n = zeros(Int, ns); flag = ones(Int, ns)
    gaps = fill(Inf, ns); lams = fill(Inf, ns)
    cM = zeros(ComplexF64, T); cM[end] = -1                             # M = -L.terms[end].coeff  (Householder.jl:92)
    upd(c) = householder_update([factorial(i - 1) * c[i] for i in 1:length(c)])
    while true
        act = [s for s in 1:ns if flag[s] == 1 && abs(z[s] - z0[s]) > tol && n[s] < maxiter]
        isempty(act) && break
        cA = Matrix{ComplexF64}(undef, T, length(act)); sig = Float64[]
        L.active = [L.eigval]; L.mode = :all
        for (q, s) in enumerate(act)
            z0[s] = z[s]
            L.params[L.eigval] = z[s]; L.params[L.auxval] = 0
            cA[:, q] = coefficients(L, z[s])
            push!(sig, (isfinite(gaps[s]) && lams[s] < 1e-4 * gaps[s]) ? 1e-5 * gaps[s] : 0.0)
        end
        local right, left
        try
            right = eigs_many_slots(fam, cA, cM, SV, act, OP_N, sig, SXR)
            left = eigs_many_slots(fam, cA, cM, SW, act, OP_C, sig, SXL)
        catch excp
            for s in act; flag[s] = excp isa LinearAlgebra.SingularException ? -6 : -2; end
            break
        end
        moved = Int[]
        for (q, s) in enumerate(act)
            if right[q] isa EigsError || left[q] isa EigsError; flag[s] = -4; continue; end
            lam_r, gap = right[q]
            isfinite(gap) && (gaps[s] = gap)
            lams[s] = abs(lam_r)
            L.params[L.eigval] = z[s]; L.params[L.auxval] = lam_r
            local dz
            try
                dz = upd(eigval_series_slots(fam, L.auxval, L.eigval, order, SXR, s, SXL, s))
            catch excp
                flag[s] = excp isa LinearAlgebra.SingularException ? -6 : -2
                continue
            end
            lam[s] = lam_r
            output && println(s, " ", n[s], "\t", abs(lam[s]), "\t", abs(dz), "\t", z[s])
            z[s] += relax * dz
            push!(moved, s)
            n[s] += 1
        end
        if !isempty(moved)                                              # v0 = (1 - relax) v0 + relax v  (Householder.jl:173-176), on the device
            slot_axpby(fam, SV, moved, SXR, moved, relax, 1 - relax)
            slot_axpby(fam, SW, moved, SXL, moved, relax, 1 - relax)
        end
    end
    # Householder.jl:189-190 for all start values at once: two batched forms, two scalings
    nv = slot_forms(fam, repeat(cM, 1, ns), SV, allc, SV, allc)
    slot_axpby(fam, SV, allc, SV, allc, 1.0 ./ sqrt.(nv), 0.0)
    cD = Matrix{ComplexF64}(undef, T, ns)
    saved = copy(L.params); L.active = [L.eigval]; L.mode = :all
    for s in 1:ns
        L.params[L.eigval] = z[s]; L.params[L.auxval] = isfinite(lam[s]) ? lam[s] : 0
        cD[:, s] = coefficients(L, z[s], 1)
    end
    merge!(L.params, saved); L.active, L.mode = active, mode
    dw = slot_forms(fam, cD, SW, allc, SV, allc)
    slot_axpby(fam, SW, allc, SW, allc, 1.0 ./ conj.(dw), 0.0)
    V = slot_read(fam, SV, 1, ns); W = slot_read(fam, SW, 1, ns)
    out = Tuple{Solution,Int,Int}[]
    for s in 1:ns
        f = flag[s]
        L.params[L.eigval] = z[s]; L.params[L.auxval] = isfinite(lam[s]) ? lam[s] : 0
        if f == 1
            f = n[s] >= maxiter ? -1 : (abs(lam[s]) <= lam_tol ? 1 : (abs(z[s] - z0[s]) <= tol ? 0 : (isnan(z[s]) ? -5 : -3)))
        end
        push!(out, (Solution(L.params, V[:, s], W[:, s], L.eigval), n[s], f))
    end
    L.active, L.mode = active, mode
    return out
end

"`solve` (solver.jl:36-184) with the local refinement of ALL new Beyn estimates of a cycle in one lock-step batch
(`householder_many`, which also delivers the adjoint vectors the deflation of the moments needs, solver.jl:131-137)"
function solve_batched(fam::DeviceFamily, Γ; Δl=1, N=16, tol=1e-8, eigvals=Dict(), max_outer_cycles=1, atol_σ=1e-12, rtol_σ=1e-8, order=1, loglevel=0)
    L = fam.L
    d = size(L.terms[1].coeff, 1)
    eigvals = Dict{ComplexF64,Any}(eigvals)
    A = Array{Array{ComplexF64,3},1}()
    l = 0
    for cycle in 1:max_outer_cycles * max(1, div(d, Δl))
        l >= d && break
        V = zeros(ComplexF64, d, Δl); for i in 1:Δl; l + i <= d && (V[l+i, i] = 1); end
        mom = compute_moment_matrices(fam, Γ, V; K=1, N=N)
        for (ω, val) in eigvals                                        # deflation of known eigenpairs (solver.jl:57-64,131-137)
            sol = val[1]; val[2] || continue
            for p in 0:size(mom, 3)-1
                mom[:, :, p+1] .-= (-2π * im * ω^p) .* (sol.v * (sol.v_adj[l+1:l+Δl])')
            end
        end
        push!(A, mom); l += Δl
        B0 = hcat((a[:, :, 1] for a in A)...); B1 = hcat((a[:, :, 2] for a in A)...)
        F = svd(B0)
        σmax = isempty(F.S) ? 0.0 : F.S[1]
        keep = (F.S .> atol_σ) .& (F.S .> rtol_σ * σmax)
        any(keep) || break
        U, S, W = F.U[:, keep], F.S[keep], F.V[:, keep]
        Ω, P = eigen(U' * B1 * W * Diagonal(1 ./ S)); P = U * P
        inside = [j for j in 1:length(Ω) if inpoly(Ω[j], Γ)]
        found_new = false
        for (sol, n, flag) in householder_many(fam, Ω[inside]; maxiter=10, tol=tol, order=order, v0s=P[:, inside])
            flag >= 0 || continue
            ω = sol.params[L.eigval]
            any(abs(ω - w) <= 10tol * max(1, abs(w)) for w in keys(eigvals)) && continue
            eigvals[ω] = Any[sol, inpoly(ω, Γ)]
            found_new = true
            loglevel > 0 && println("solve_batched: new eigenvalue ", ω, " after ", n, " Newton steps")
        end
        (!found_new && sum(keep) < l) && break                         # rank gap reached and nothing new: done (solver.jl:172)
    end
    return eigvals
end

# ---------------------------------------------------------------------------------------------------------------
# the snapshot basis of a handle on the host (exchange between GPUs when the caller runs its own process group instead of
# compute_moment_matrices(::Vector{DeviceFamily}, ...) below)
# ---------------------------------------------------------------------------------------------------------------
"(S, l, kact, Hk, g) = rb_export(fam): the projected terms of the handle's snapshot basis (wae_rb_export): Hk[c, i, s, ki] =
q_iᴴ A_k q_s of column c's basis for the terms kact (0-based term indices), g[c, i] = q_iᴴ v_c"
function rb_export(fam::DeviceFamily)
    S = Ref{Int32}(0); l = Ref{Int32}(0); nk = Ref{Int32}(0)
    check(ccall((:wae_rb_export, libwaehip), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ptr{Int32}, Ptr{ComplexF64}, Ptr{ComplexF64}),
                fam.handle, S, l, nk, C_NULL, C_NULL, C_NULL))
    kact = zeros(Int32, nk[]); Hk = zeros(ComplexF64, l[], S[], S[], nk[]); g = zeros(ComplexF64, l[], S[])
    check(ccall((:wae_rb_export, libwaehip), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ptr{Int32}, Ptr{ComplexF64}, Ptr{ComplexF64}),
                fam.handle, S, l, nk, kact, Hk, g))
    return Int(S[]), Int(l[]), kact, Hk, g
end

"install a basis whose vectors lie at the device address Q_dev (S x d x l, interleaved [row][column]) for mode 2 (wae_rb_import)"
function rb_import(fam::DeviceFamily, Q_dev::UInt64, kact::Vector{Int32}, Hk::Array{ComplexF64,4}, g::Matrix{ComplexF64})
    l, S = size(g)
    check(ccall((:wae_rb_import, libwaehip), Cint, (Ptr{Cvoid}, Int32, Int32, UInt64, Int32, Ptr{Int32}, Ptr{ComplexF64}, Ptr{ComplexF64}),
                fam.handle, S, l, Q_dev, length(kact), kact, Hk, g))
    return
end

# ---------------------------------------------------------------------------------------------------------------
# P1 assembly and shape sensitivity on the device (SURVEY 8f-2, 8f-4): the element loops of `discretize` for P1 meshes
# (Helmholtz.jl:405-487) and the local re-discretisations of `discrete_adjoint_shape_sensitivity` (shape_sensitivity.jl:16-141).
# points: 3 x N Float64 (`mesh.points`); tets / tris: 4 x ntet / 3 x ntri matrices of 1-BASED point indices
# (`hcat(mesh.tetrahedra...)`); the wrappers hand 0-based Int32 copies to the library.
# ---------------------------------------------------------------------------------------------------------------
_zero_based(a) = Matrix{Int32}(a) .- Int32(1)

function _take_p1(h::Ptr{Cvoid}, both::Bool)
    try
        n = Ref{Int64}(0); nz = Ref{Int64}(0)
        check(ccall((:wae_p1_info, libwaehip), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), h, n, nz))
        rowptr = zeros(Int32, n[] + 1); col = zeros(Int32, nz[]); m = zeros(Float64, nz[]); k = both ? zeros(Float64, nz[]) : Float64[]
        check(ccall((:wae_p1_get, libwaehip), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}), h, rowptr, col, m, both ? k : C_NULL))
        # the arrays are CSR: as (colptr, rowval) they describe the TRANSPOSE; transpose back (M, K, C are symmetric, Q is not)
        csr(v) = SparseMatrixCSC{ComplexF64,UInt32}(sparse(transpose(SparseMatrixCSC(Int(n[]), Int(n[]), UInt32.(rowptr .+ 1), UInt32.(col .+ 1), ComplexF64.(v)))))
        return both ? (csr(m), csr(k)) : csr(m)
    finally
        ccall((:wae_p1_free, libwaehip), Cint, (Ptr{Cvoid},), h)
    end
end

"M, K = assemble_p1(points, tets; c_tet, device): mass and stiffness (K = -c² ∫∇φ_a·∇φ_b, Helmholtz.jl:120-124,405-441) on the device"
function assemble_p1(points::Matrix{Float64}, tets::AbstractMatrix{<:Integer}; c_tet=nothing, device::Integer=0)
    t0 = _zero_based(tets); h = Ref{Ptr{Cvoid}}(C_NULL)
    cc = c_tet === nothing ? C_NULL : Vector{Float64}(c_tet)
    check(ccall((:wae_p1_assemble, libwaehip), Cint, (Int32, Int64, Ptr{Float64}, Int64, Ptr{Int32}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                device, size(points, 2), points, size(t0, 2), t0, cc, h))
    return _take_p1(h[], true)
end

"C = assemble_p1_boundary(points, tris; c_tri, device): admittance-boundary operator -i c |e1×e2| (1+δ_ab)/24 (Helmholtz.jl:443-463)"
function assemble_p1_boundary(points::Matrix{Float64}, tris::AbstractMatrix{<:Integer}; c_tri=nothing, device::Integer=0)
    t0 = _zero_based(tris); h = Ref{Ptr{Cvoid}}(C_NULL)
    cc = c_tri === nothing ? C_NULL : Vector{Float64}(c_tri)
    check(ccall((:wae_p1_assemble_boundary, libwaehip), Cint, (Int32, Int64, Ptr{Float64}, Int64, Ptr{Int32}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                device, size(points, 2), points, size(t0, 2), t0, cc, h))
    return -1im .* _take_p1(h[], false)
end

"Q, V_flame = assemble_p1_flame(points, tets, flame_tets, ref_tet, n_ref, nglobal_scaled; device): Q = Σ_flame S ⊗ g
(Helmholtz.jl:292-344,464-487); flame_tets / ref_tet are 1-based indices into tets; nglobal_scaled = (γ-1)/ρ·Q02U0"
function assemble_p1_flame(points::Matrix{Float64}, tets::AbstractMatrix{<:Integer}, flame_tets::AbstractVector{<:Integer}, ref_tet::Integer,
                           n_ref::Vector{Float64}, nglobal_scaled::Real; device::Integer=0)
    t0 = _zero_based(tets); fl = Vector{Int32}(flame_tets) .- Int32(1); h = Ref{Ptr{Cvoid}}(C_NULL); vol = Ref{Float64}(0.0)
    check(ccall((:wae_p1_assemble_flame, libwaehip), Cint,
                (Int32, Int64, Ptr{Float64}, Int64, Ptr{Int32}, Int64, Ptr{Int32}, Int32, Ptr{Float64}, Float64, Ref{Ptr{Cvoid}}, Ref{Float64}),
                device, size(points, 2), points, size(t0, 2), t0, length(fl), fl, ref_tet - 1, n_ref, Float64(nglobal_scaled), h, vol))
    return _take_p1(h[], false), vol[]
end

"sens (3 x length(surface_points)) = discrete_adjoint_shape_sensitivity_p1(points, tets, c_tet, surface_points, ω, v, v_adj; ...):
-v_adjᴴ (dL/dx) v of shape_sensitivity.jl:16-141 for the interior operators and, if bnd_tris is given, the admittance boundary ω·Y·C;
v, v_adj normalised as there (vᴴv = 1, v_adjᴴ L'(ω) v = 1).  One device thread per (point, adjacent simplex, coordinate); the pairs of a
point are summed here in order."
function discrete_adjoint_shape_sensitivity_p1(points::Matrix{Float64}, tets::AbstractMatrix{<:Integer}, c_tet, surface_points::AbstractVector{<:Integer},
                                               ω::ComplexF64, v::Vector{ComplexF64}, v_adj::Vector{ComplexF64};
                                               bnd_tris=nothing, bnd_c=nothing, Y=0.0, h::Float64=1e-9, device::Integer=0)
    t0 = _zero_based(tets)
    lut = zeros(Int, size(points, 2)); for (i, p) in enumerate(surface_points); lut[p] = i; end
    pair_pt_t = Int32[]; pair_tet = Int32[]; own_t = Int[]
    for e in 1:size(t0, 2), a in 1:4
        p = t0[a, e] + 1
        lut[p] > 0 && (push!(pair_pt_t, p - 1); push!(pair_tet, e - 1); push!(own_t, lut[p]))
    end
    s0 = bnd_tris === nothing ? zeros(Int32, 3, 0) : _zero_based(bnd_tris)
    pair_pt_s = Int32[]; pair_tri = Int32[]; own_s = Int[]
    for e in 1:size(s0, 2), a in 1:3
        p = s0[a, e] + 1
        lut[p] > 0 && (push!(pair_pt_s, p - 1); push!(pair_tri, e - 1); push!(own_s, lut[p]))
    end
    out_t = zeros(ComplexF64, 3, length(pair_tet)); out_s = zeros(ComplexF64, 3, length(pair_tri))
    cc = c_tet === nothing ? C_NULL : Vector{Float64}(c_tet)
    bc = (bnd_c === nothing || isempty(pair_tri)) ? C_NULL : Vector{Float64}(bnd_c)
    om = Float64[real(ω), imag(ω)]; omy = Float64[real(ω * Y), imag(ω * Y)]
    check(ccall((:wae_p1_shape_sensitivity, libwaehip), Cint,
                (Int32, Int64, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Int64, Ptr{Int32}, Ptr{Int32},
                 Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Float64, Ptr{ComplexF64}, Ptr{ComplexF64}),
                device, size(points, 2), points, t0, cc, length(pair_tet), pair_pt_t, pair_tet, isempty(pair_tri) ? C_NULL : s0, bc, length(pair_tri),
                isempty(pair_tri) ? C_NULL : pair_pt_s, isempty(pair_tri) ? C_NULL : pair_tri, size(t0, 2), size(s0, 2), om, omy, v, v_adj, h,
                isempty(pair_tet) ? C_NULL : out_t, isempty(pair_tri) ? C_NULL : out_s))
    sens = zeros(ComplexF64, 3, length(surface_points))
    for (q, i) in enumerate(own_t); sens[:, i] .+= out_t[:, q]; end
    for (q, i) in enumerate(own_s); sens[:, i] .+= out_s[:, q]; end
    return sens
end

"flame part of the same sensitivity (a :flame entry in dscrp; shape_sensitivity.jl:62-141): -v_adjᴴ coeff (Q₊ - Q₋)/(2h) v per surface
point and coordinate, Q re-discretised on the flame domain REDUCED to the tetrahedra at the point (its volume included, Helmholtz.jl:325);
coeff = the flame term's scalar n·exp(-iωτ) at ω.  flame_tets / ref_tet 1-based."
function discrete_adjoint_shape_sensitivity_p1_flame(points::Matrix{Float64}, tets::AbstractMatrix{<:Integer}, surface_points::AbstractVector{<:Integer},
                                                     flame_tets::AbstractVector{<:Integer}, ref_tet::Integer, n_ref::Vector{Float64}, nglobal_scaled::Real,
                                                     coeff::ComplexF64, v::Vector{ComplexF64}, v_adj::Vector{ComplexF64}; h::Float64=1e-9, device::Integer=0)
    t0 = _zero_based(tets)
    ns = length(surface_points)
    lut = zeros(Int, size(points, 2)); for (i, p) in enumerate(surface_points); lut[p] = i; end
    pair_pt = Int32[]; pair_tet = Int32[]; own = Int[]
    for e in flame_tets, a in 1:4
        p = t0[a, e] + 1
        lut[p] > 0 && (push!(pair_pt, p - 1); push!(pair_tet, e - 1); push!(own, lut[p]))
    end
    pair_pt_r = Int32[]; own_r = Int[]
    for a in 1:4
        p = t0[a, ref_tet] + 1
        lut[p] > 0 && (push!(pair_pt_r, p - 1); push!(own_r, lut[p]))
    end
    np, nr = length(pair_tet), length(pair_pt_r)
    det_pm = zeros(Float64, 2, 3, np); ssum = zeros(ComplexF64, np); g_pm = zeros(ComplexF64, 2, 3, nr); g0 = zeros(ComplexF64, 1)
    check(ccall((:wae_p1_shape_sensitivity_flame, libwaehip), Cint,
                (Int32, Int64, Ptr{Float64}, Int64, Ptr{Int32}, Int64, Ptr{Int32}, Ptr{Int32}, Int32, Int64, Ptr{Int32}, Ptr{Float64},
                 Ptr{ComplexF64}, Ptr{ComplexF64}, Float64, Ptr{Float64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}),
                device, size(points, 2), points, size(t0, 2), t0, np, np == 0 ? C_NULL : pair_pt, np == 0 ? C_NULL : pair_tet, ref_tet - 1, nr,
                nr == 0 ? C_NULL : pair_pt_r, n_ref, v, v_adj, h, np == 0 ? C_NULL : det_pm, np == 0 ? C_NULL : ssum, nr == 0 ? C_NULL : g_pm, g0))
    a_pm = zeros(ComplexF64, 2, 3, ns); V_pm = zeros(Float64, 2, 3, ns)      # v_adjᴴ S± and the volume of the point's flame tetrahedra
    for (q, i) in enumerate(own)
        a_pm[:, :, i] .+= det_pm[:, :, q] ./ 24 .* ssum[q]; V_pm[:, :, i] .+= det_pm[:, :, q] ./ 6
    end
    b_pm = fill(g0[1], 2, 3, ns)                                             # Σ_b ∇φ_b·n_ref v_b on the reference tetrahedron
    for (q, i) in enumerate(own_r); b_pm[:, :, i] = g_pm[:, :, q]; end
    out = zeros(ComplexF64, 3, ns)
    for i in 1:ns
        V_pm[1, 1, i] > 0 || continue                                        # no flame tetrahedron at the point: empty domain, no term
        q = a_pm[:, :, i] .* (-(Float64(nglobal_scaled) ./ V_pm[:, :, i]) .* b_pm[:, :, i])      # v_adjᴴ Q± v  (g = -nlocal ∇φ·n_ref)
        out[:, i] = -coeff .* (q[1, :] .- q[2, :]) ./ (2h)
    end
    return out
end

# ---------------------------------------------------------------------------------------------------------------
# several GPUs of one node from this one Julia process (SURVEY.md 8b/8e): one DeviceFamily replica per device
# ---------------------------------------------------------------------------------------------------------------
"moments like `compute_moment_matrices`, quadrature points and snapshot phase shared out over `fams` (one replica per GPU:
`[DeviceFamily(L; device=g) for g in 0:7]`); RCCL all-gather / reduce inside the library (wae_beyn_moments_mgpu)"
function compute_moment_matrices(fams::Vector{DeviceFamily}, Γ, V::Matrix{ComplexF64}; K=1, N=16, rb=nothing)
    foreach(ensure_solver!, fams)
    L = fams[1].L
    X, W = FastGaussQuadrature.gausslegendre(N)
    zs = ComplexF64[]; ws = ComplexF64[]
    for i in 1:length(Γ)
        a, b = Γ[i], Γ[i == length(Γ) ? 1 : i + 1]
        append!(zs, X .* (b - a) / 2 .+ (a + b) / 2); append!(ws, W .* (b - a) / 2)
    end
    saved = (L.active, L.mode); L.active = [L.eigval]; L.mode = :all
    ct = Matrix{ComplexF64}(undef, length(L.terms), length(zs))
    for (j, z) in enumerate(zs); ct[:, j] = coefficients(L, z); end
    L.active, L.mode = saved
    d, l = size(V); npts = length(zs)
    rb === nothing && (rb = (npts >= 64 && d >= 1000) ? min(40, div(npts, 2)) : 0)
    A = zeros(ComplexF64, d, l, 2K); info = Ref{SolveInfo}()
    hs = [f.handle for f in fams]
    code = check(ccall((:wae_beyn_moments_mgpu, libwaehip), Cint,
                (Ptr{Ptr{Cvoid}}, Int32, Int32, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32, Int32, Float64, Int32,
                 Int32, Ptr{ComplexF64}, Ref{SolveInfo}),
                hs, length(hs), npts, zs, ws, ct, V, l, K, fams[1].tol, fams[1].maxit, rb, A, info))
    report(code, info[], "beyn moments ($(length(hs)) GPUs)"; fatal=true)
    return A
end

# ---------------------------------------------------------------------------------------------------------------
# operator interchange: the WAEFAM1 container read by wae_amd.nlevp.save.load_family (Python harness).  The text format
# of `save(fname, L)` (LinOpFam.jl:236-294) also loads there, but a 1M-DoF family is ~1 GB of text; this writes the
# CSC arrays as they lie in memory.  Coefficient functions are stored by name: `names` maps closures (e.g. the
# `exp_plus` made in Helmholtz.jl:90) to a constructor expression such as "generate_exp_az(0.0+0.19634954084936207im)".
# ---------------------------------------------------------------------------------------------------------------
_jsonstr(s::AbstractString) = "\"" * replace(replace(String(s), "\\" => "\\\\"), "\"" => "\\\"") * "\""
_jsonnum(x::Real) = isnan(x) ? "NaN" : (isinf(x) ? (x > 0 ? "Infinity" : "-Infinity") : string(Float64(x)))

function save_family_bin(fname::AbstractString, L::LinearOperatorFamily; names=Dict{Any,String}())
    io = IOBuffer()
    print(io, "{\"version\": 1, \"eigval\": ", _jsonstr(string(L.eigval)), ", \"auxval\": ", _jsonstr(string(L.auxval)),
          ", \"active\": [", join((_jsonstr(string(a)) for a in L.active), ", "), "], \"mode\": ", _jsonstr(string(L.mode)),
          ", \"params\": {")
    print(io, join((_jsonstr(string(k)) * ": [" * _jsonnum(real(v)) * ", " * _jsonnum(imag(v)) * "]" for (k, v) in L.params), ", "))
    print(io, "}, \"terms\": [")
    for (i, t) in enumerate(L.terms)
        i > 1 && print(io, ", ")
        fn = [haskey(names, f) ? names[f] : string(nameof(f)) for f in t.func]
        m, n = size(t.coeff)
        print(io, "{\"symbol\": ", _jsonstr(t.symbol), ", \"operator\": ", _jsonstr(t.operator), ", \"functions\": [",
              join((_jsonstr(f) for f in fn), ", "), "], \"params\": [",
              join(("[" * join((_jsonstr(string(q)) for q in p), ", ") * "]" for p in t.params), ", "),
              "], \"m\": ", m, ", \"n\": ", n, ", \"nnz\": ", nnz(t.coeff), ", \"base\": 1}")
    end
    print(io, "]}")
    head = take!(io)
    open(fname, "w") do f
        write(f, "WAEFAM1\n")
        write(f, UInt64(length(head)))
        write(f, head)
        pad() = write(f, zeros(UInt8, mod(-position(f), 8)))
        for t in L.terms
            A = t.coeff
            pad(); write(f, Vector{Int64}(A.colptr))
            pad(); write(f, Vector{Int64}(A.rowval))
            pad(); write(f, Vector{ComplexF64}(A.nzval))
        end
    end
    return fname
end

end # module
