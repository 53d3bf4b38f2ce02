// P1 tetrahedral assembly of the Helmholtz mass and stiffness matrices on the device.
//
// Replaces the element loops of `discretize` for the two operators that dominate its run time
// (src/Helmholtz.jl:405-441: every tetrahedron contributes a 4x4 block to M and to K) with
//   1. one kernel: per tetrahedron the coordinate transformation (src/FEM/FEM.jl:9-20), the local mass matrix
//      |det J|/120 (1 + delta_ab) (FEM.jl:704-710) and the local stiffness matrix -c^2 |det J|/6 grad phi_a . grad phi_b
//      (FEM.jl:1745-1766, Helmholtz.jl:120-124), written as 16 (key = row*np + col, m, k) triplets;
//   2. a stable radix sort of the keys (hipCUB) -- duplicates become adjacent, in element order, so the sums below are
//      deterministic (no atomics);
//   3. reduce-by-key (hipCUB) of both value streams, and the CSR row pointer from the unique keys.
// The result is what Julia's sparse(I, J, V) returns for the same triplets, up to the order of the additions.
#include <hipcub/hipcub.hpp>

#include <cstring>
#include <memory>
#include <vector>

#include "wae_internal.h"
#include "../../include/waehip.h"

namespace {

template <class F> int wae_guarded(F &&f) {
    try {
        return f();
    } catch (const WaeError &e) {
        wae_set_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        wae_set_error(e.what());
        return WAE_ERR_INVALID;
    }
}

// local P1 matrices of one tetrahedron with corner coordinates X: M_ab = |det J|/120 (1+delta_ab), K_ab = -c^2 |det J|/6 grad_a.grad_b
__device__ inline void p1_local(const double X[4][3], double c, double M[16], double K[16]) {
    double J[3][3];
    for (int r = 0; r < 3; ++r)
        for (int a = 0; a < 3; ++a) J[r][a] = X[a][r] - X[3][r];
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double adet = fabs(det);
    const double id = 1.0 / det;
    double G[4][3];
    G[0][0] = c00 * id; G[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; G[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    G[1][0] = c01 * id; G[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; G[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    G[2][0] = c02 * id; G[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; G[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    for (int k = 0; k < 3; ++k) G[3][k] = -(G[0][k] + G[1][k] + G[2][k]);
    const double ks = -(c * c) * adet / 6.0;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            M[a * 4 + b] = adet * (a == b ? 2.0 : 1.0) / 120.0;
            K[a * 4 + b] = ks * (G[a][0] * G[b][0] + G[a][1] * G[b][1] + G[a][2] * G[b][2]);
        }
}

// Discrete-adjoint shape sensitivity (src/shape_sensitivity.jl:16-141), interior part: for the pair (surface point p,
// adjacent tetrahedron t) and coordinate x:  out = -v_adj_loc^H [ w^2 (M+ - M-) + (K+ - K-) ] v_loc / (2h), M+-/K+- the local
// matrices with x_p moved by +-h (central difference of two local re-discretisations, as the reference does).
__global__ __launch_bounds__(256) void shape_tet_kernel(const double *__restrict__ pts, const int *__restrict__ tets, const double *__restrict__ c_tet,
                                                        int64_t npair, const int *__restrict__ pair_pt, const int *__restrict__ pair_tet,
                                                        double wr, double wi, const cplx *__restrict__ v, const cplx *__restrict__ vadj, double h,
                                                        cplx *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= npair * 3) return;
    const int64_t pr = e / 3;
    const int crd = (int)(e - pr * 3);
    const int t = pair_tet[pr], p = pair_pt[pr];
    int vtx[4];
    double X[4][3];
    int a0 = -1;
    for (int a = 0; a < 4; ++a) {
        vtx[a] = tets[(size_t)t * 4 + a];
        if (vtx[a] == p) a0 = a;
        for (int k = 0; k < 3; ++k) X[a][k] = pts[(size_t)vtx[a] * 3 + k];
    }
    if (a0 < 0) { out[e] = cplx{0.0, 0.0}; return; }
    const double c = c_tet ? c_tet[t] : 1.0;
    const double x0 = X[a0][crd];
    double Mp[16], Kp[16], Mm[16], Km[16];
    X[a0][crd] = x0 + h; p1_local(X, c, Mp, Kp);
    X[a0][crd] = x0 - h; p1_local(X, c, Mm, Km);
    const double w2r = wr * wr - wi * wi, w2i = 2.0 * wr * wi;
    const double s = 1.0 / (2.0 * h);
    cplx acc = {0.0, 0.0};
    for (int a = 0; a < 4; ++a) {
        const cplx ya = vadj[vtx[a]];
        for (int b = 0; b < 4; ++b) {
            const double dm = (Mp[a * 4 + b] - Mm[a * 4 + b]) * s, dk = (Kp[a * 4 + b] - Km[a * 4 + b]) * s;
            const cplx D = {w2r * dm + dk, w2i * dm};
            const cplx xb = v[vtx[b]];
            const cplx Dx = {D.x * xb.x - D.y * xb.y, D.x * xb.y + D.y * xb.x};
            acc.x += ya.x * Dx.x + ya.y * Dx.y;          // conj(ya) * Dx
            acc.y += ya.x * Dx.y - ya.y * Dx.x;
        }
    }
    out[e] = cplx{-acc.x, -acc.y};
}

// boundary (admittance) part: C_ab = -i c |(x0-x2) x (x1-x2)| (1+delta_ab)/24 (FEM.jl:435-441, Helmholtz.jl:151-156,459),
// operator term w*Y*C:  out = -v_adj_loc^H [ w Y (C+ - C-) ] v_loc / (2h)
__global__ __launch_bounds__(256) void shape_tri_kernel(const double *__restrict__ pts, const int *__restrict__ tris, const double *__restrict__ c_tri,
                                                        int64_t npair, const int *__restrict__ pair_pt, const int *__restrict__ pair_tri,
                                                        double wyr, double wyi, const cplx *__restrict__ v, const cplx *__restrict__ vadj, double h,
                                                        cplx *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= npair * 3) return;
    const int64_t pr = e / 3;
    const int crd = (int)(e - pr * 3);
    const int t = pair_tri[pr], p = pair_pt[pr];
    int vtx[3];
    double X[3][3];
    int a0 = -1;
    for (int a = 0; a < 3; ++a) {
        vtx[a] = tris[(size_t)t * 3 + a];
        if (vtx[a] == p) a0 = a;
        for (int k = 0; k < 3; ++k) X[a][k] = pts[(size_t)vtx[a] * 3 + k];
    }
    if (a0 < 0) { out[e] = cplx{0.0, 0.0}; return; }
    auto area2 = [&]() {
        const double u0 = X[0][0] - X[2][0], u1 = X[0][1] - X[2][1], u2 = X[0][2] - X[2][2];
        const double w0 = X[1][0] - X[2][0], w1 = X[1][1] - X[2][1], w2 = X[1][2] - X[2][2];
        const double n0 = u1 * w2 - u2 * w1, n1 = u2 * w0 - u0 * w2, n2 = u0 * w1 - u1 * w0;
        return sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    };
    const double x0 = X[a0][crd];
    X[a0][crd] = x0 + h; const double dp = area2();
    X[a0][crd] = x0 - h; const double dm = area2();
    const double dd = c_tri[t] * (dp - dm) / (2.0 * h) / 24.0;          // d/dx of c |..| /24; C = -i * that * (1+delta)
    // w Y C' = (wyr + i wyi) * (-i) * dd * (1+delta) = (wyi - i wyr) dd (1+delta)
    const cplx f = {wyi * dd, -wyr * dd};
    cplx acc = {0.0, 0.0};
    for (int a = 0; a < 3; ++a) {
        const cplx ya = vadj[vtx[a]];
        for (int b = 0; b < 3; ++b) {
            const double m = (a == b) ? 2.0 : 1.0;
            const cplx xb = v[vtx[b]];
            const cplx Dx = {m * (f.x * xb.x - f.y * xb.y), m * (f.x * xb.y + f.y * xb.x)};
            acc.x += ya.x * Dx.x + ya.y * Dx.y;
            acc.y += ya.x * Dx.y - ya.y * Dx.x;
        }
    }
    out[e] = cplx{-acc.x, -acc.y};
}

__global__ __launch_bounds__(256) void p1_local_kernel(const double *__restrict__ pts, const int *__restrict__ tets, const double *__restrict__ c_tet,
                                                       int64_t nt, int64_t np, unsigned long long *__restrict__ keys, double *__restrict__ mv,
                                                       double *__restrict__ kv) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nt) return;
    int v[4];
    double X[4][3];
    for (int a = 0; a < 4; ++a) {
        v[a] = tets[t * 4 + a];
        for (int k = 0; k < 3; ++k) X[a][k] = pts[(size_t)v[a] * 3 + k];
    }
    double Ml[16], Kl[16];
    p1_local(X, c_tet ? c_tet[t] : 1.0, Ml, Kl);
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            const size_t o = (size_t)t * 16 + a * 4 + b;
            keys[o] = (unsigned long long)v[a] * (unsigned long long)np + (unsigned long long)v[b];
            mv[o] = Ml[a * 4 + b];
            kv[o] = Kl[a * 4 + b];
        }
}

__global__ __launch_bounds__(256) void gather_d_kernel(const double *__restrict__ src, const unsigned int *__restrict__ idx, double *__restrict__ dst, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) dst[e] = src[idx[e]];
}
__global__ __launch_bounds__(256) void iota_kernel(unsigned int *__restrict__ idx, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) idx[e] = (unsigned int)e;
}
// col[e] = key % np; rows counted into rowcnt[key / np + 1]
__global__ __launch_bounds__(256) void split_keys_kernel(const unsigned long long *__restrict__ keys, size_t nnz, unsigned long long np,
                                                         int *__restrict__ col, int *__restrict__ rowptr) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nnz; e += (size_t)gridDim.x * 256) {
        const unsigned long long k = keys[e];
        const unsigned long long r = k / np;
        col[e] = (int)(k - r * np);
        // first entry of a row records its position; rows without entries are filled by the host scan
        if (e == 0 || keys[e - 1] / np != r) rowptr[r] = (int)e;
    }
}

struct P1Handle {
    int64_t np = 0, nnz = 0;
    std::vector<int> rowptr, col;
    std::vector<double> m, k;
};

template <class T> struct Dev {
    T *p = nullptr;
    explicit Dev(size_t n) { HIP_CHECK(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T))); }
    ~Dev() { if (p) (void)hipFree(p); }
};

// keyed triplets (key = row * np + col; up to two value streams) -> CSR: stable radix sort (duplicates adjacent, in
// element order: the sums are deterministic, no atomics), reduce-by-key, row pointer from the unique keys
P1Handle *triplets_to_csr(int64_t npoints, size_t ne, Dev<unsigned long long> &k0, Dev<double> &mv, Dev<double> *kv) {
    Dev<double> ms(ne), ks(kv ? ne : 1), mu(ne), ku(kv ? ne : 1);
    Dev<int> dcol(ne), drow((size_t)npoints + 1), dnum(1);
    Dev<unsigned long long> k1(ne), ku0(ne);
    Dev<unsigned int> i0(ne), i1(ne);
    const unsigned g = (unsigned)std::min<size_t>((ne + 255) / 256, 8192);
    hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(256), 0, 0, i0.p, ne);
    int bits = 1;
    while (bits < 64 && ((unsigned long long)npoints * (unsigned long long)npoints) >> bits) ++bits;
    size_t tmp_bytes = 0;
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0.p, k1.p, i0.p, i1.p, (int)ne, 0, bits));
    Dev<char> tmp(tmp_bytes);
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, k0.p, k1.p, i0.p, i1.p, (int)ne, 0, bits));
    hipLaunchKernelGGL(gather_d_kernel, dim3(g), dim3(256), 0, 0, mv.p, i1.p, ms.p, ne);
    if (kv) hipLaunchKernelGGL(gather_d_kernel, dim3(g), dim3(256), 0, 0, kv->p, i1.p, ks.p, ne);
    size_t tb2 = 0;
    HIP_CHECK(hipcub::DeviceReduce::ReduceByKey(nullptr, tb2, k1.p, ku0.p, ms.p, mu.p, dnum.p, hipcub::Sum(), (int)ne));
    Dev<char> tmp2(tb2);
    HIP_CHECK(hipcub::DeviceReduce::ReduceByKey(tmp2.p, tb2, k1.p, ku0.p, ms.p, mu.p, dnum.p, hipcub::Sum(), (int)ne));
    if (kv) HIP_CHECK(hipcub::DeviceReduce::ReduceByKey(tmp2.p, tb2, k1.p, ku0.p, ks.p, ku.p, dnum.p, hipcub::Sum(), (int)ne));
    int nnz = 0;
    HIP_CHECK(hipMemcpy(&nnz, dnum.p, sizeof(int), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemset(drow.p, 0xff, ((size_t)npoints + 1) * sizeof(int)));          // -1 = row without entries
    hipLaunchKernelGGL(split_keys_kernel, dim3(g), dim3(256), 0, 0, ku0.p, (size_t)nnz, (unsigned long long)npoints, dcol.p, drow.p);
    HIP_CHECK(hipGetLastError());
    std::unique_ptr<P1Handle> H(new P1Handle);
    H->np = npoints; H->nnz = nnz;
    H->rowptr.resize((size_t)npoints + 1); H->col.resize(nnz); H->m.resize(nnz); H->k.assign(nnz, 0.0);
    HIP_CHECK(hipMemcpy(H->rowptr.data(), drow.p, ((size_t)npoints + 1) * sizeof(int), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(H->col.data(), dcol.p, (size_t)nnz * sizeof(int), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(H->m.data(), mu.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost));
    if (kv) HIP_CHECK(hipMemcpy(H->k.data(), ku.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost));
    H->rowptr[npoints] = nnz;
    for (int64_t r = npoints - 1; r >= 0; --r)
        if (H->rowptr[r] < 0) H->rowptr[r] = H->rowptr[r + 1];
    return H.release();
}

// boundary mass of one triangle: b_ab = c |(x0-x2) x (x1-x2)| (1+delta_ab)/24  (FEM.jl:9-20,435-441; Helmholtz.jl:151-156); C = -i b
__global__ __launch_bounds__(256) void p1_boundary_kernel(const double *__restrict__ pts, const int *__restrict__ tris, const double *__restrict__ c_tri,
                                                          int64_t nt, int64_t np, unsigned long long *__restrict__ keys, double *__restrict__ bv) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nt) return;
    int v[3];
    double X[3][3];
    for (int a = 0; a < 3; ++a) {
        v[a] = tris[t * 3 + a];
        for (int k = 0; k < 3; ++k) X[a][k] = pts[(size_t)v[a] * 3 + k];
    }
    const double u0 = X[0][0] - X[2][0], u1 = X[0][1] - X[2][1], u2 = X[0][2] - X[2][2];
    const double w0 = X[1][0] - X[2][0], w1 = X[1][1] - X[2][1], w2 = X[1][2] - X[2][2];
    const double n0 = u1 * w2 - u2 * w1, n1 = u2 * w0 - u0 * w2, n2 = u0 * w1 - u1 * w0;
    const double det = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    const double c = c_tri ? c_tri[t] : 1.0;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            const size_t o = (size_t)t * 9 + a * 3 + b;
            keys[o] = (unsigned long long)v[a] * (unsigned long long)np + (unsigned long long)v[b];
            bv[o] = c * ((a == b ? 2.0 : 1.0) / 24.0) * det;
        }
}

// |det J| of the listed tetrahedra (volume source S_a = |det J|/24 per node, FEM.jl:2429-2431; flame volume = sum |det J|/6)
__global__ __launch_bounds__(256) void p1_det_kernel(const double *__restrict__ pts, const int *__restrict__ tets, const int *__restrict__ list, int64_t n,
                                                     double *__restrict__ adet) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int t = list[i];
    double X[4][3];
    for (int a = 0; a < 4; ++a)
        for (int k = 0; k < 3; ++k) X[a][k] = pts[(size_t)tets[(size_t)t * 4 + a] * 3 + k];
    double J[3][3];
    for (int r = 0; r < 3; ++r)
        for (int a = 0; a < 3; ++a) J[r][a] = X[a][r] - X[3][r];
    const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) + J[0][1] * (J[1][2] * J[2][0] - J[1][0] * J[2][2]) +
                       J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
    adet[i] = fabs(det);
}
// Q triplets: (node a of flame tetrahedron i, node b of the reference tetrahedron) -> |det J_i|/24 * g_b   (Helmholtz.jl:19-33,483)
__global__ __launch_bounds__(256) void p1_flame_kernel(const int *__restrict__ tets, const int *__restrict__ list, int64_t n, const double *__restrict__ adet,
                                                       int r0, int r1, int r2, int r3, double g0, double g1, double g2, double g3, int64_t np,
                                                       unsigned long long *__restrict__ keys, double *__restrict__ qv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int t = list[i];
    const double s = adet[i] / 24.0;
    const int rn[4] = {r0, r1, r2, r3};
    const double g[4] = {g0, g1, g2, g3};
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            const size_t o = (size_t)i * 16 + a * 4 + b;
            keys[o] = (unsigned long long)tets[(size_t)t * 4 + a] * (unsigned long long)np + (unsigned long long)rn[b];
            qv[o] = s * g[b];
        }
}

}  // namespace

// Flame part of the discrete-adjoint shape sensitivity (shape_sensitivity.jl:62-141 with a :flame domain in dscrp).  The
// reference re-discretises, per surface point, the flame tetrahedra that touch the point: Q = S (x) g with S_a = |det J|/24 on
// their nodes (FEM.jl:2429-2431), nlocal = nglobal_scaled / (volume of THOSE tetrahedra) (Helmholtz.jl:325 on the reduced domain)
// and g_b = -nlocal grad(phi_b).n_ref on the reference tetrahedron (FEM.jl:2442-2448).  Per (point, flame tetrahedron,
// coordinate) this kernel returns |det J| with the point moved by +h and by -h, and per pair the sum of conj(v_adj) over the
// tetrahedron's nodes: the host sums them per point (fixed order) and forms  -v_adj' (Q+ - Q-)/(2h) v.
__global__ __launch_bounds__(256) void shape_flame_kernel(const double *__restrict__ pts, const int *__restrict__ tets, int64_t npair,
                                                          const int *__restrict__ pair_pt, const int *__restrict__ pair_tet,
                                                          const cplx *__restrict__ vadj, double h, double *__restrict__ det_pm,
                                                          cplx *__restrict__ ssum) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= npair * 3) return;
    const int64_t pr = e / 3;
    const int crd = (int)(e - pr * 3);
    const int t = pair_tet[pr], p = pair_pt[pr];
    int vtx[4];
    double X[4][3];
    int a0 = -1;
    cplx sa = {0.0, 0.0};
    for (int a = 0; a < 4; ++a) {
        vtx[a] = tets[(size_t)t * 4 + a];
        if (vtx[a] == p) a0 = a;
        for (int k = 0; k < 3; ++k) X[a][k] = pts[(size_t)vtx[a] * 3 + k];
        const cplx y = vadj[vtx[a]];
        sa.x += y.x; sa.y -= y.y;                              // conj(v_adj)
    }
    if (crd == 0) ssum[pr] = sa;
    auto adet = [&]() {
        double J[3][3];
        for (int r = 0; r < 3; ++r)
            for (int a = 0; a < 3; ++a) J[r][a] = X[a][r] - X[3][r];
        return fabs(J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) + J[0][1] * (J[1][2] * J[2][0] - J[1][0] * J[2][2]) +
                    J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]));
    };
    if (a0 < 0) { det_pm[e * 2] = det_pm[e * 2 + 1] = adet(); return; }
    const double x0 = X[a0][crd];
    X[a0][crd] = x0 + h; det_pm[e * 2] = adet();
    X[a0][crd] = x0 - h; det_pm[e * 2 + 1] = adet();
}
// sum_b (grad(phi_b) . n_ref) v_b on the reference tetrahedron with vertex `p` moved by +h / -h along each coordinate (3 x 2
// complex numbers per listed vertex), and undisplaced (g0)
__global__ void shape_ref_kernel(const double *__restrict__ pts, const int *__restrict__ tets, int ref_tet, int64_t npair, const int *__restrict__ pair_pt,
                                 double n0, double n1, double n2, const cplx *__restrict__ v, double h, cplx *__restrict__ g_pm, cplx *__restrict__ g0) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e > npair * 6) return;                                  // e == npair * 6: the undisplaced value
    int vtx[4];
    double X[4][3];
    for (int a = 0; a < 4; ++a) {
        vtx[a] = tets[(size_t)ref_tet * 4 + a];
        for (int k = 0; k < 3; ++k) X[a][k] = pts[(size_t)vtx[a] * 3 + k];
    }
    if (e < npair * 6) {
        const int64_t pr = e / 6;
        const int crd = (int)((e - pr * 6) >> 1), sgn = (int)(e & 1);
        for (int a = 0; a < 4; ++a)
            if (vtx[a] == pair_pt[pr]) X[a][crd] += sgn ? -h : h;
    }
    double J[3][3];
    for (int r = 0; r < 3; ++r)
        for (int a = 0; a < 3; ++a) J[r][a] = X[a][r] - X[3][r];
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double id = 1.0 / (J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02);
    double G[4][3];
    G[0][0] = c00 * id; G[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; G[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    G[1][0] = c01 * id; G[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; G[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    G[2][0] = c02 * id; G[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; G[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    for (int k = 0; k < 3; ++k) G[3][k] = -(G[0][k] + G[1][k] + G[2][k]);
    cplx acc = {0.0, 0.0};
    for (int b = 0; b < 4; ++b) {
        const double gb = G[b][0] * n0 + G[b][1] * n1 + G[b][2] * n2;
        const cplx xb = v[vtx[b]];
        acc.x += gb * xb.x; acc.y += gb * xb.y;
    }
    if (e < npair * 6) g_pm[e] = acc; else *g0 = acc;
}

extern "C" {

int wae_p1_assemble_boundary(int32_t device, int64_t npoints, const double *points, int64_t ntris, const int32_t *tris, const double *c_tri, void **out) {
    return wae_guarded([&]() {
        if (!(npoints > 0 && ntris > 0 && points && tris && out)) throw WaeError(WAE_ERR_INVALID, "bad argument");
        for (int64_t i = 0; i < ntris * 3; ++i)
            if (tris[i] < 0 || tris[i] >= npoints) throw WaeError(WAE_ERR_INVALID, "triangle refers to a point outside 0..npoints-1");
        HIP_CHECK(hipSetDevice(device));
        const size_t ne = (size_t)ntris * 9;
        Dev<double> dpts((size_t)npoints * 3), dc(c_tri ? (size_t)ntris : 1), bv(ne);
        Dev<int> dt((size_t)ntris * 3);
        Dev<unsigned long long> k0(ne);
        HIP_CHECK(hipMemcpy(dpts.p, points, (size_t)npoints * 3 * sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dt.p, tris, (size_t)ntris * 3 * sizeof(int), hipMemcpyHostToDevice));
        if (c_tri) HIP_CHECK(hipMemcpy(dc.p, c_tri, (size_t)ntris * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(p1_boundary_kernel, dim3((unsigned)((ntris + 255) / 256)), dim3(256), 0, 0, dpts.p, dt.p, c_tri ? dc.p : nullptr, ntris, npoints,
                           k0.p, bv.p);
        HIP_CHECK(hipGetLastError());
        *out = triplets_to_csr(npoints, ne, k0, bv, nullptr);
        return WAE_OK;
    });
}

int wae_p1_assemble_flame(int32_t device, int64_t npoints, const double *points, int64_t ntets, const int32_t *tets, int64_t nflame,
                          const int32_t *flame_tets, int32_t ref_tet, const double *n_ref, double nglobal_scaled, void **out, double *volume_out) {
    return wae_guarded([&]() {
        if (!(npoints > 0 && ntets > 0 && nflame > 0 && points && tets && flame_tets && n_ref && out)) throw WaeError(WAE_ERR_INVALID, "bad argument");
        if (ref_tet < 0 || ref_tet >= ntets) throw WaeError(WAE_ERR_INVALID, "reference tetrahedron out of range");
        for (int64_t i = 0; i < nflame; ++i)
            if (flame_tets[i] < 0 || flame_tets[i] >= ntets) throw WaeError(WAE_ERR_INVALID, "flame tetrahedron out of range");
        for (int64_t i = 0; i < ntets * 4; ++i)
            if (tets[i] < 0 || tets[i] >= npoints) throw WaeError(WAE_ERR_INVALID, "tetrahedron refers to a point outside 0..npoints-1");
        HIP_CHECK(hipSetDevice(device));
        const size_t ne = (size_t)nflame * 16;
        Dev<double> dpts((size_t)npoints * 3), adet((size_t)nflame), vol(1), qv(ne);
        Dev<int> dt((size_t)ntets * 4), dl((size_t)nflame);
        Dev<unsigned long long> k0(ne);
        HIP_CHECK(hipMemcpy(dpts.p, points, (size_t)npoints * 3 * sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dt.p, tets, (size_t)ntets * 4 * sizeof(int), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dl.p, flame_tets, (size_t)nflame * sizeof(int), hipMemcpyHostToDevice));
        const unsigned gb = (unsigned)((nflame + 255) / 256);
        hipLaunchKernelGGL(p1_det_kernel, dim3(gb), dim3(256), 0, 0, dpts.p, dt.p, dl.p, nflame, adet.p);
        HIP_CHECK(hipGetLastError());
        // flame volume = sum |det J| / 6 (Meshutils.jl:757-767), summed in a fixed tree order on the device
        size_t tb = 0;
        HIP_CHECK(hipcub::DeviceReduce::Sum(nullptr, tb, adet.p, vol.p, (int)nflame));
        Dev<char> tmp(tb);
        HIP_CHECK(hipcub::DeviceReduce::Sum(tmp.p, tb, adet.p, vol.p, (int)nflame));
        double det_sum = 0.0;
        HIP_CHECK(hipMemcpy(&det_sum, vol.p, sizeof(double), hipMemcpyDeviceToHost));
        const double volume = det_sum / 6.0;
        if (!(volume > 0.0)) throw WaeError(WAE_ERR_INVALID, "flame domain has no volume");
        if (volume_out) *volume_out = volume;
        const double nlocal = nglobal_scaled / volume;                                 // Helmholtz.jl:325
        // g_b = -nlocal grad(phi_b) . n_ref on the reference tetrahedron (FEM.jl:2442-2448, Helmholtz.jl:482): 4 numbers, on the host
        int rn[4];
        double X[4][3];
        for (int a = 0; a < 4; ++a) {
            rn[a] = tets[(size_t)ref_tet * 4 + a];
            for (int k = 0; k < 3; ++k) X[a][k] = points[(size_t)rn[a] * 3 + k];
        }
        double J[3][3];
        for (int r = 0; r < 3; ++r)
            for (int a = 0; a < 3; ++a) J[r][a] = X[a][r] - X[3][r];
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
        if (det == 0.0) throw WaeError(WAE_ERR_INVALID, "degenerate reference tetrahedron");
        const double id = 1.0 / det;
        double G[4][3];
        G[0][0] = c00 * id; G[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; G[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
        G[1][0] = c01 * id; G[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; G[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
        G[2][0] = c02 * id; G[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; G[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
        for (int k = 0; k < 3; ++k) G[3][k] = -(G[0][k] + G[1][k] + G[2][k]);
        double g[4];
        for (int b = 0; b < 4; ++b) g[b] = -nlocal * (G[b][0] * n_ref[0] + G[b][1] * n_ref[1] + G[b][2] * n_ref[2]);
        hipLaunchKernelGGL(p1_flame_kernel, dim3(gb), dim3(256), 0, 0, dt.p, dl.p, nflame, adet.p, rn[0], rn[1], rn[2], rn[3], g[0], g[1], g[2], g[3], npoints,
                           k0.p, qv.p);
        HIP_CHECK(hipGetLastError());
        *out = triplets_to_csr(npoints, ne, k0, qv, nullptr);
        return WAE_OK;
    });
}

int wae_p1_assemble(int32_t device, int64_t npoints, const double *points, int64_t ntets, const int32_t *tets, const double *c_tet, void **out) {
    return wae_guarded([&]() {
        if (!(npoints > 0 && ntets > 0 && points && tets && out)) throw WaeError(WAE_ERR_INVALID, "bad argument");
        if ((size_t)ntets * 16 >= 0xffffffffull) throw WaeError(WAE_ERR_INVALID, "too many tetrahedra for 32-bit triplet indices");
        for (int64_t i = 0; i < ntets * 4; ++i)
            if (tets[i] < 0 || tets[i] >= npoints) throw WaeError(WAE_ERR_INVALID, "tetrahedron refers to a point outside 0..npoints-1");
        HIP_CHECK(hipSetDevice(device));
        const size_t ne = (size_t)ntets * 16;
        Dev<double> dpts((size_t)npoints * 3), dc(c_tet ? (size_t)ntets : 1), mv(ne), kv(ne);
        Dev<int> dt((size_t)ntets * 4);
        Dev<unsigned long long> k0(ne);
        HIP_CHECK(hipMemcpy(dpts.p, points, (size_t)npoints * 3 * sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dt.p, tets, (size_t)ntets * 4 * sizeof(int), hipMemcpyHostToDevice));
        if (c_tet) HIP_CHECK(hipMemcpy(dc.p, c_tet, (size_t)ntets * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(p1_local_kernel, dim3((unsigned)((ntets + 255) / 256)), dim3(256), 0, 0, dpts.p, dt.p, c_tet ? dc.p : nullptr, ntets, npoints,
                           k0.p, mv.p, kv.p);
        HIP_CHECK(hipGetLastError());
        *out = triplets_to_csr(npoints, ne, k0, mv, &kv);
        return WAE_OK;
    });
}

int wae_p1_shape_sensitivity(int32_t device, int64_t npoints, const double *points, const int32_t *tets, const double *c_tet, int64_t npair_t,
                             const int32_t *pair_pt_t, const int32_t *pair_tet, const int32_t *tris, const double *c_tri, int64_t npair_s,
                             const int32_t *pair_pt_s, const int32_t *pair_tri, int64_t ntets, int64_t ntris, const double *omega,
                             const double *omegaY, const double *v, const double *v_adj, double h, double *out_t, double *out_s) {
    return wae_guarded([&]() {
        if (!(npoints > 0 && points && v && v_adj && omega && h > 0.0)) throw WaeError(WAE_ERR_INVALID, "bad argument");
        if (npair_t > 0 && !(tets && pair_pt_t && pair_tet && out_t && ntets > 0)) throw WaeError(WAE_ERR_INVALID, "bad tetrahedron pair arguments");
        if (npair_s > 0 && !(tris && c_tri && pair_pt_s && pair_tri && out_s && omegaY && ntris > 0)) throw WaeError(WAE_ERR_INVALID, "bad triangle pair arguments");
        for (int64_t i = 0; i < npair_t; ++i)
            if (pair_tet[i] < 0 || pair_tet[i] >= ntets || pair_pt_t[i] < 0 || pair_pt_t[i] >= npoints) throw WaeError(WAE_ERR_INVALID, "pair index out of range");
        for (int64_t i = 0; i < npair_s; ++i)
            if (pair_tri[i] < 0 || pair_tri[i] >= ntris || pair_pt_s[i] < 0 || pair_pt_s[i] >= npoints) throw WaeError(WAE_ERR_INVALID, "pair index out of range");
        for (int64_t i = 0; i < ntets * 4 && npair_t > 0; ++i)
            if (tets[i] < 0 || tets[i] >= npoints) throw WaeError(WAE_ERR_INVALID, "tetrahedron refers to a point outside 0..npoints-1");
        for (int64_t i = 0; i < ntris * 3 && npair_s > 0; ++i)
            if (tris[i] < 0 || tris[i] >= npoints) throw WaeError(WAE_ERR_INVALID, "triangle refers to a point outside 0..npoints-1");
        HIP_CHECK(hipSetDevice(device));
        Dev<double> dpts((size_t)npoints * 3);
        Dev<cplx> dv((size_t)npoints), dva((size_t)npoints);
        HIP_CHECK(hipMemcpy(dpts.p, points, (size_t)npoints * 3 * sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dv.p, v, (size_t)npoints * sizeof(cplx), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dva.p, v_adj, (size_t)npoints * sizeof(cplx), hipMemcpyHostToDevice));
        if (npair_t > 0) {
            Dev<int> dt((size_t)ntets * 4), dpp((size_t)npair_t), dpt((size_t)npair_t);
            Dev<double> dc(c_tet ? (size_t)ntets : 1);
            Dev<cplx> dout((size_t)npair_t * 3);
            HIP_CHECK(hipMemcpy(dt.p, tets, (size_t)ntets * 4 * sizeof(int), hipMemcpyHostToDevice));
            if (c_tet) HIP_CHECK(hipMemcpy(dc.p, c_tet, (size_t)ntets * sizeof(double), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(dpp.p, pair_pt_t, (size_t)npair_t * sizeof(int), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(dpt.p, pair_tet, (size_t)npair_t * sizeof(int), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(shape_tet_kernel, dim3((unsigned)((npair_t * 3 + 255) / 256)), dim3(256), 0, 0, dpts.p, dt.p, c_tet ? dc.p : nullptr, npair_t,
                               dpp.p, dpt.p, omega[0], omega[1], dv.p, dva.p, h, dout.p);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpy(out_t, dout.p, (size_t)npair_t * 3 * sizeof(cplx), hipMemcpyDeviceToHost));
        }
        if (npair_s > 0) {
            Dev<int> dt((size_t)ntris * 3), dpp((size_t)npair_s), dpt((size_t)npair_s);
            Dev<double> dc((size_t)ntris);
            Dev<cplx> dout((size_t)npair_s * 3);
            HIP_CHECK(hipMemcpy(dt.p, tris, (size_t)ntris * 3 * sizeof(int), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(dc.p, c_tri, (size_t)ntris * sizeof(double), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(dpp.p, pair_pt_s, (size_t)npair_s * sizeof(int), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(dpt.p, pair_tri, (size_t)npair_s * sizeof(int), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(shape_tri_kernel, dim3((unsigned)((npair_s * 3 + 255) / 256)), dim3(256), 0, 0, dpts.p, dt.p, dc.p, npair_s, dpp.p, dpt.p,
                               omegaY[0], omegaY[1], dv.p, dva.p, h, dout.p);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpy(out_s, dout.p, (size_t)npair_s * 3 * sizeof(cplx), hipMemcpyDeviceToHost));
        }
        return WAE_OK;
    });
}

int wae_p1_shape_sensitivity_flame(int32_t device, int64_t npoints, const double *points, int64_t ntets, const int32_t *tets, int64_t npair,
                                   const int32_t *pair_pt, const int32_t *pair_tet, int32_t ref_tet, int64_t npair_r, const int32_t *pair_pt_r,
                                   const double *n_ref, const double *v, const double *v_adj, double h, double *det_pm, double *ssum,
                                   double *g_pm, double *g0) {
    return wae_guarded([&]() {
        if (!(npoints > 0 && ntets > 0 && points && tets && n_ref && v && v_adj && h > 0.0 && g0)) throw WaeError(WAE_ERR_INVALID, "bad argument");
        if (npair > 0 && !(pair_pt && pair_tet && det_pm && ssum)) throw WaeError(WAE_ERR_INVALID, "bad flame pair arguments");
        if (npair_r > 0 && !(pair_pt_r && g_pm)) throw WaeError(WAE_ERR_INVALID, "bad reference pair arguments");
        if (ref_tet < 0 || ref_tet >= ntets) throw WaeError(WAE_ERR_INVALID, "reference tetrahedron out of range");
        for (int64_t i = 0; i < npair; ++i)
            if (pair_tet[i] < 0 || pair_tet[i] >= ntets || pair_pt[i] < 0 || pair_pt[i] >= npoints) throw WaeError(WAE_ERR_INVALID, "pair index out of range");
        for (int64_t i = 0; i < npair_r; ++i)
            if (pair_pt_r[i] < 0 || pair_pt_r[i] >= npoints) throw WaeError(WAE_ERR_INVALID, "pair index out of range");
        for (int64_t i = 0; i < ntets * 4; ++i)
            if (tets[i] < 0 || tets[i] >= npoints) throw WaeError(WAE_ERR_INVALID, "tetrahedron refers to a point outside 0..npoints-1");
        HIP_CHECK(hipSetDevice(device));
        Dev<double> dpts((size_t)npoints * 3);
        Dev<cplx> dv((size_t)npoints), dva((size_t)npoints);
        Dev<int> dt((size_t)ntets * 4);
        HIP_CHECK(hipMemcpy(dpts.p, points, (size_t)npoints * 3 * sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dv.p, v, (size_t)npoints * sizeof(cplx), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dva.p, v_adj, (size_t)npoints * sizeof(cplx), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dt.p, tets, (size_t)ntets * 4 * sizeof(int), hipMemcpyHostToDevice));
        if (npair > 0) {
            Dev<int> dpp((size_t)npair), dpt((size_t)npair);
            Dev<double> ddet((size_t)npair * 6);
            Dev<cplx> dss((size_t)npair);
            HIP_CHECK(hipMemcpy(dpp.p, pair_pt, (size_t)npair * sizeof(int), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(dpt.p, pair_tet, (size_t)npair * sizeof(int), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(shape_flame_kernel, dim3((unsigned)((npair * 3 + 255) / 256)), dim3(256), 0, 0, dpts.p, dt.p, npair, dpp.p, dpt.p, dva.p, h,
                               ddet.p, dss.p);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpy(det_pm, ddet.p, (size_t)npair * 6 * sizeof(double), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(ssum, dss.p, (size_t)npair * sizeof(cplx), hipMemcpyDeviceToHost));
        }
        {
            Dev<int> dpr((size_t)std::max<int64_t>(npair_r, 1));
            Dev<cplx> dg((size_t)npair_r * 6 + 1);
            if (npair_r > 0) HIP_CHECK(hipMemcpy(dpr.p, pair_pt_r, (size_t)npair_r * sizeof(int), hipMemcpyHostToDevice));
            const int64_t nthr = npair_r * 6 + 1;
            hipLaunchKernelGGL(shape_ref_kernel, dim3((unsigned)((nthr + 63) / 64)), dim3(64), 0, 0, dpts.p, dt.p, ref_tet, npair_r, dpr.p, n_ref[0], n_ref[1],
                               n_ref[2], dv.p, h, dg.p, dg.p + (size_t)npair_r * 6);
            HIP_CHECK(hipGetLastError());
            if (npair_r > 0) HIP_CHECK(hipMemcpy(g_pm, dg.p, (size_t)npair_r * 6 * sizeof(cplx), hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(g0, dg.p + (size_t)npair_r * 6, sizeof(cplx), hipMemcpyDeviceToHost));
        }
        return WAE_OK;
    });
}

int wae_p1_info(const void *handle, int64_t *npoints, int64_t *nnz) {
    return wae_guarded([&]() {
        if (!handle) throw WaeError(WAE_ERR_INVALID, "null handle");
        const P1Handle *H = (const P1Handle *)handle;
        if (npoints) *npoints = H->np;
        if (nnz) *nnz = H->nnz;
        return WAE_OK;
    });
}

int wae_p1_get(const void *handle, int32_t *rowptr, int32_t *col, double *mass, double *stiff) {
    return wae_guarded([&]() {
        if (!handle) throw WaeError(WAE_ERR_INVALID, "null handle");
        const P1Handle *H = (const P1Handle *)handle;
        if (rowptr) memcpy(rowptr, H->rowptr.data(), H->rowptr.size() * sizeof(int));
        if (col) memcpy(col, H->col.data(), H->col.size() * sizeof(int));
        if (mass) memcpy(mass, H->m.data(), H->m.size() * sizeof(double));
        if (stiff) memcpy(stiff, H->k.data(), H->k.size() * sizeof(double));
        return WAE_OK;
    });
}

int wae_p1_free(void *handle) {
    delete (P1Handle *)handle;
    return WAE_OK;
}

}  // extern "C"
