// Layout probe for one MI355X: does the way a tile kernel's windows lie in memory matter to the memory system?  A multivector of n rows
// x 64 complex columns is read (and, in the second form, copied) tile by tile and chunk by chunk, exactly the bytes of one operator
// product: a workgroup draws tiles of 512 consecutive rows and walks their 8 chunks of 8 columns (128 B per row and chunk).
//   interleaved : X[row][64]  -- a chunk of a row is 128 B out of a 1-KB row (what the library has): stride 1 KB
//   panel       : X[chunk][row][8] -- a chunk of a tile is 64 KB contiguous
// Same total bytes (1 GiB read, + 1 GiB written in the copy forms); workgroups of 512 threads, one per CU x 8 as the tile kernel.  dev/ only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool PANEL, bool COPY>
__global__ __launch_bounds__(512) void k_tiles(const double2 *__restrict__ x, double2 *__restrict__ y, double *out, int ntiles, size_t n) {
    double s = 0.0;
    const int t8 = threadIdx.x & 7, r0 = threadIdx.x >> 3;            // 8 lanes cover the 128 B of a (row, chunk); 64 rows per pass
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t row0 = (size_t)t * 512;
        for (int c = 0; c < 8; ++c) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const size_t row = row0 + r0 + 64 * p;
                const size_t e = PANEL ? ((size_t)c * n + row) * 8 + t8 : row * 64 + c * 8 + t8;
                const double2 v = x[e];
                if (COPY) y[e] = double2{v.x + 1.0, v.y};
                else s += v.x + v.y;
            }
        }
    }
    if (!COPY && s == 1.2345e300) out[0] = s;
}

int main() {
    const size_t n = (size_t)1 << 20;                  // 1M rows x 64 columns x 16 B = 1 GiB
    const size_t n2 = n * 64;
    double2 *a, *b; double *out;
    CK(hipMalloc(&a, n2 * 16)); CK(hipMalloc(&b, n2 * 16)); CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 0, n2 * 16)); CK(hipMemset(b, 0, n2 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ntiles = (int)(n / 512), reps = 10;
    auto timeit = [&](const char *name, double bytes, int grid, auto launch) {
        for (int i = 0; i < 2; ++i) launch(grid);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch(grid);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-22s grid %5d : %7.1f us  %5.2f TB/s\n", name, grid, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    };
    for (int grid : {256, 512, 1024, 2048}) {
        timeit("read  interleaved", n2 * 16.0, grid, [&](int g) { hipLaunchKernelGGL((k_tiles<false, false>), dim3(g), dim3(512), 0, 0, a, b, out, ntiles, n); });
        timeit("read  panel", n2 * 16.0, grid, [&](int g) { hipLaunchKernelGGL((k_tiles<true, false>), dim3(g), dim3(512), 0, 0, a, b, out, ntiles, n); });
        timeit("copy  interleaved", 2 * n2 * 16.0, grid, [&](int g) { hipLaunchKernelGGL((k_tiles<false, true>), dim3(g), dim3(512), 0, 0, a, b, out, ntiles, n); });
        timeit("copy  panel", 2 * n2 * 16.0, grid, [&](int g) { hipLaunchKernelGGL((k_tiles<true, true>), dim3(g), dim3(512), 0, 0, a, b, out, ntiles, n); });
    }
    return 0;
}
