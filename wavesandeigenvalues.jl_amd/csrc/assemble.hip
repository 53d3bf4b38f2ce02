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
    // J columns x_a - x_4, a = 1..3
    double J[3][3];
    for (int r = 0; r < 3; ++r)
        for (int a = 0; a < 3; ++a) J[r][a] = X[a][r] - X[3][r];
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
    const double adet = fabs(det);
    const double id = 1.0 / det;
    // inverse of J (rows of Jinv = gradients of the first three barycentric coordinates)
    double G[4][3];
    G[0][0] = c00 * id; G[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; G[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    G[1][0] = c01 * id; G[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; G[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    G[2][0] = c02 * id; G[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; G[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    for (int k = 0; k < 3; ++k) G[3][k] = -(G[0][k] + G[1][k] + G[2][k]);
    const double c = c_tet ? c_tet[t] : 1.0;
    const double ks = -(c * c) * adet / 6.0;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            const size_t o = (size_t)t * 16 + a * 4 + b;
            keys[o] = (unsigned long long)v[a] * (unsigned long long)np + (unsigned long long)v[b];
            mv[o] = adet * (a == b ? 2.0 : 1.0) / 120.0;
            kv[o] = ks * (G[a][0] * G[b][0] + G[a][1] * G[b][1] + G[a][2] * G[b][2]);
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

}  // namespace

extern "C" {

int wae_p1_assemble(int32_t device, int64_t npoints, const double *points, int64_t ntets, const int32_t *tets, const double *c_tet, void **out) {
    return wae_guarded([&]() {
        if (!(npoints > 0 && ntets > 0 && points && tets && out)) throw WaeError(WAE_ERR_INVALID, "bad argument");
        if ((size_t)ntets * 16 >= 0xffffffffull) throw WaeError(WAE_ERR_INVALID, "too many tetrahedra for 32-bit triplet indices");
        for (int64_t i = 0; i < ntets * 4; ++i)
            if (tets[i] < 0 || tets[i] >= npoints) throw WaeError(WAE_ERR_INVALID, "tetrahedron refers to a point outside 0..npoints-1");
        HIP_CHECK(hipSetDevice(device));
        const size_t ne = (size_t)ntets * 16;
        Dev<double> dpts((size_t)npoints * 3), dc(c_tet ? (size_t)ntets : 1), mv(ne), kv(ne), ms(ne), ks(ne), mu(ne), ku(ne);
        Dev<int> dt((size_t)ntets * 4), dcol(ne), drow((size_t)npoints + 1), dnum(1);
        Dev<unsigned long long> k0(ne), k1(ne), ku0(ne);
        Dev<unsigned int> i0(ne), i1(ne);
        HIP_CHECK(hipMemcpy(dpts.p, points, (size_t)npoints * 3 * sizeof(double), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dt.p, tets, (size_t)ntets * 4 * sizeof(int), hipMemcpyHostToDevice));
        if (c_tet) HIP_CHECK(hipMemcpy(dc.p, c_tet, (size_t)ntets * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(p1_local_kernel, dim3((unsigned)((ntets + 255) / 256)), dim3(256), 0, 0, dpts.p, dt.p, c_tet ? dc.p : nullptr, ntets, npoints,
                           k0.p, mv.p, kv.p);
        HIP_CHECK(hipGetLastError());
        const unsigned g = (unsigned)std::min<size_t>((ne + 255) / 256, 8192);
        hipLaunchKernelGGL(iota_kernel, dim3(g), dim3(256), 0, 0, i0.p, ne);
        // stable sort of (key, triplet index)
        int bits = 1;
        while (bits < 64 && ((unsigned long long)npoints * (unsigned long long)npoints) >> bits) ++bits;
        size_t tmp_bytes = 0;
        HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k0.p, k1.p, i0.p, i1.p, (int)ne, 0, bits));
        Dev<char> tmp(tmp_bytes);
        HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, k0.p, k1.p, i0.p, i1.p, (int)ne, 0, bits));
        hipLaunchKernelGGL(gather_d_kernel, dim3(g), dim3(256), 0, 0, mv.p, i1.p, ms.p, ne);
        hipLaunchKernelGGL(gather_d_kernel, dim3(g), dim3(256), 0, 0, kv.p, i1.p, ks.p, ne);
        // segment sums
        size_t tb2 = 0;
        HIP_CHECK(hipcub::DeviceReduce::ReduceByKey(nullptr, tb2, k1.p, ku0.p, ms.p, mu.p, dnum.p, hipcub::Sum(), (int)ne));
        Dev<char> tmp2(tb2);
        HIP_CHECK(hipcub::DeviceReduce::ReduceByKey(tmp2.p, tb2, k1.p, ku0.p, ms.p, mu.p, dnum.p, hipcub::Sum(), (int)ne));
        HIP_CHECK(hipcub::DeviceReduce::ReduceByKey(tmp2.p, tb2, k1.p, ku0.p, ks.p, ku.p, dnum.p, hipcub::Sum(), (int)ne));
        int nnz = 0;
        HIP_CHECK(hipMemcpy(&nnz, dnum.p, sizeof(int), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemset(drow.p, 0xff, ((size_t)npoints + 1) * sizeof(int)));          // -1 = row without entries
        hipLaunchKernelGGL(split_keys_kernel, dim3(g), dim3(256), 0, 0, ku0.p, (size_t)nnz, (unsigned long long)npoints, dcol.p, drow.p);
        HIP_CHECK(hipGetLastError());
        auto *H = new P1Handle;
        H->np = npoints; H->nnz = nnz;
        H->rowptr.resize((size_t)npoints + 1); H->col.resize(nnz); H->m.resize(nnz); H->k.resize(nnz);
        HIP_CHECK(hipMemcpy(H->rowptr.data(), drow.p, ((size_t)npoints + 1) * sizeof(int), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(H->col.data(), dcol.p, (size_t)nnz * sizeof(int), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(H->m.data(), mu.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(H->k.data(), ku.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost));
        H->rowptr[npoints] = nnz;
        for (int64_t r = npoints - 1; r >= 0; --r)
            if (H->rowptr[r] < 0) H->rowptr[r] = H->rowptr[r + 1];
        *out = H;
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
