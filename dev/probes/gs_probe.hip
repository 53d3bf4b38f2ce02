// Streaming-rate probe for the Gram-Schmidt kernels' access pattern: nv + 1 interleaved multivectors (d x 64 complex each) read row-group by
// row-group, one partial sum per vector -- what dots_kernel does -- by grid size and rows per thread and iteration.  dev/ only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double2 cplx;
template <int NV, int RPT>
__global__ __launch_bounds__(256) void k_dots(const cplx *__restrict__ V, size_t stride, const cplx *__restrict__ W, long n, cplx *__restrict__ partial) {
    const int nb = 64, R = 4;
    const int tid = threadIdx.x, b = tid % nb, rl = tid / nb;
    cplx acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = cplx{0.0, 0.0};
    for (long row = (long)blockIdx.x * R * RPT + rl; row < n; row += (long)gridDim.x * R * RPT) {
        cplx w[RPT], v[RPT][NV];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const long r2 = row + q * R < n ? row + q * R : n - 1;
            const size_t e = (size_t)r2 * nb + b;
            w[q] = W[e];
#pragma unroll
            for (int i = 0; i < NV; ++i) v[q][i] = V[(size_t)i * stride + e];
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q)
#pragma unroll
            for (int i = 0; i < NV; ++i) { acc[i].x += v[q][i].x * w[q].x + v[q][i].y * w[q].y; acc[i].y += v[q][i].x * w[q].y - v[q][i].y * w[q].x; }
    }
    cplx s = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NV; ++i) { s.x += acc[i].x; s.y += acc[i].y; }
    if (s.x == 1.2345e300) partial[blockIdx.x] = s;
}
// contiguous block ranges instead of a grid stride
template <int NV>
__global__ __launch_bounds__(256) void k_dots_blk(const cplx *__restrict__ V, size_t stride, const cplx *__restrict__ W, long n, cplx *__restrict__ partial) {
    const int nb = 64, R = 4;
    const int tid = threadIdx.x, b = tid % nb, rl = tid / nb;
    cplx acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = cplx{0.0, 0.0};
    const long per = ((n + gridDim.x - 1) / gridDim.x + R - 1) / R * R;
    const long lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for (long row = lo + rl; row < hi; row += R) {
        const size_t e = (size_t)row * nb + b;
        const cplx w = W[e];
#pragma unroll
        for (int i = 0; i < NV; ++i) { const cplx v = V[(size_t)i * stride + e]; acc[i].x += v.x * w.x + v.y * w.y; acc[i].y += v.x * w.y - v.y * w.x; }
    }
    cplx s = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NV; ++i) { s.x += acc[i].x; s.y += acc[i].y; }
    if (s.x == 1.2345e300) partial[blockIdx.x] = s;
}
int main() {
    const long n = 995328; const int nb = 64;
    const size_t vec = (size_t)n * nb;
    const int NVMAX = 32;
    cplx *V, *W, *P;
    CK(hipMalloc(&V, vec * 16 * NVMAX)); CK(hipMalloc(&W, vec * 16)); CK(hipMalloc(&P, 1 << 20));
    CK(hipMemset(V, 0, vec * 16 * NVMAX)); CK(hipMemset(W, 0, vec * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, int nv, int grid, auto launch) {
        for (int i = 0; i < 2; ++i) launch(grid);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; ++i) launch(grid);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-10s nv %2d grid %5d : %8.1f us  %5.2f TB/s\n", name, nv, grid, ms / 5 * 1e3, (double)(nv + 1) * vec * 16 * 5 / (ms * 1e-3) / 1e12);
    };
    for (int grid : {512, 768, 1024, 1536, 2048, 4096}) {
        timeit("dots", 8, grid, [&](int g) { hipLaunchKernelGGL((k_dots<8, 1>), dim3(g), dim3(256), 0, 0, V, vec, W, n, P); });
        timeit("dots_r2", 8, grid, [&](int g) { hipLaunchKernelGGL((k_dots<8, 2>), dim3(g), dim3(256), 0, 0, V, vec, W, n, P); });
        timeit("dots_blk", 8, grid, [&](int g) { hipLaunchKernelGGL((k_dots_blk<8>), dim3(g), dim3(256), 0, 0, V, vec, W, n, P); });
        timeit("dots", 16, grid, [&](int g) { hipLaunchKernelGGL((k_dots<16, 1>), dim3(g), dim3(256), 0, 0, V, vec, W, n, P); });
        timeit("dots_r2", 16, grid, [&](int g) { hipLaunchKernelGGL((k_dots<16, 2>), dim3(g), dim3(256), 0, 0, V, vec, W, n, P); });
        timeit("dots_blk", 16, grid, [&](int g) { hipLaunchKernelGGL((k_dots_blk<16>), dim3(g), dim3(256), 0, 0, V, vec, W, n, P); });
        timeit("dots", 32, grid, [&](int g) { hipLaunchKernelGGL((k_dots<32, 1>), dim3(g), dim3(256), 0, 0, V, vec, W, n, P); });
        timeit("dots_blk", 32, grid, [&](int g) { hipLaunchKernelGGL((k_dots_blk<32>), dim3(g), dim3(256), 0, 0, V, vec, W, n, P); });
    }
    return 0;
}
