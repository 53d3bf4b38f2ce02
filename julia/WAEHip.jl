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

mutable struct DeviceFamily
    L::LinearOperatorFamily
    handle::Ptr{Cvoid}
    solver_ready::Bool
    tol::Float64
    maxit::Int32
    function DeviceFamily(L::LinearOperatorFamily; device::Integer=0, tol=1e-12, maxit=400)
        T = length(L.terms)
        d = size(L.terms[1].coeff, 1)
        # Helmholtz terms are SparseMatrixCSC{ComplexF64,UInt32} (Helmholtz.jl:407-408,515): pass colptr/rowval/nzval as they are
        mats = [SparseMatrixCSC{ComplexF64,UInt32}(sparse(t.coeff)) for t in L.terms]
        ptrs = [pointer(m.colptr) for m in mats]; idxs = [pointer(m.rowval) for m in mats]; vals = [pointer(m.nzval) for m in mats]
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve mats ptrs idxs vals begin
            check(ccall((:wae_family_create, libwaehip), Cint,
                        (Ref{Ptr{Cvoid}}, Int64, Int32, Int32, Int32, Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Int32),
                        h, d, T, 4, 1, 0 #=WAE_CSC=#, ptrs, idxs, vals, device))
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

function ensure_solver!(fam::DeviceFamily; zref=nothing, opts=Float64[])
    fam.solver_ready && return
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
    return X
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
        return A
    end
    idx, rest = _snapshot_split(npts, rb)
    A1 = zeros(ComplexF64, d, l, 2K)
    sig = (Ptr{Cvoid}, Int32, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Ptr{ComplexF64}, Int32, Int32, Float64, Int32,
           Int32, Int32, Int32, UInt64, Ptr{ComplexF64}, UInt64, Int32, Int32, Int32, Ref{SolveInfo})
    # mode 0: the snapshot points (solutions kept in the handle's store); mode 2: all other points from the projection
    check(ccall((:wae_beyn_moments_rb, libwaehip), Cint, sig, fam.handle, length(idx), zs[idx], ws[idx], ct[:, idx], V, l, K,
                fam.tol, fam.maxit, 0, length(idx), 0, 0, A, 0, 0, 0, 0, info))
    # V = C_NULL: the probe matrix uploaded by the mode-0 call is still on the device (include/waehip.h)
    check(ccall((:wae_beyn_moments_rb, libwaehip), Cint, sig, fam.handle, length(rest), zs[rest], ws[rest], ct[:, rest], C_NULL, l, K,
                fam.tol, fam.maxit, 2, length(idx), 0, 0, A1, 0, 0, 0, 0, info))
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
