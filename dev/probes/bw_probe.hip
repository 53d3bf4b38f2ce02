// Streaming-rate probe for one MI355X: what this box's memory system delivers for reads, writes, copies and the triad, by grid size
// and access form -- the yardstick the operator product's L2-miss traffic (4.5 TB/s) is to be read against.  dev/ only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ a, double *out, size_t n2) {
    double s = 0.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n2; e += (size_t)gridDim.x * 256) { double2 x = a[e]; s += x.x + x.y; }
    if (s == 1.2345e300) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(double2 *__restrict__ a, size_t n2) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n2; e += (size_t)gridDim.x * 256) a[e] = double2{1.0, 2.0};
}
__global__ __launch_bounds__(256) void k_write_nt(double2 *__restrict__ a, size_t n2) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n2; e += (size_t)gridDim.x * 256) {
        __builtin_nontemporal_store(1.0, &a[e].x); __builtin_nontemporal_store(2.0, &a[e].y);
    }
}
__global__ __launch_bounds__(256) void k_copy(double2 *__restrict__ a, const double2 *__restrict__ b, size_t n2) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n2; e += (size_t)gridDim.x * 256) a[e] = b[e];
}
template <int U>
__global__ __launch_bounds__(256) void k_copy_u(double2 *__restrict__ a, const double2 *__restrict__ b, size_t n2) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; e + (U - 1) * stride < n2; e += U * stride) {
        double2 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = b[e + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) a[e + u * stride] = x[u];
    }
    for (; e < n2; e += stride) a[e] = b[e];
}
__global__ __launch_bounds__(256) void k_triad(double2 *__restrict__ a, const double2 *__restrict__ b, const double2 *__restrict__ c, double s, size_t n2) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n2; e += (size_t)gridDim.x * 256) {
        double2 x = b[e], y = c[e];
        a[e] = double2{x.x + s * y.x, x.y + s * y.y};
    }
}
// block-contiguous variants: each workgroup owns a contiguous range (like a tile kernel's output rows)
__global__ __launch_bounds__(256) void k_copy_blk(double2 *__restrict__ a, const double2 *__restrict__ b, size_t n2) {
    const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = lo + per < n2 ? lo + per : n2;
    for (size_t e = lo + threadIdx.x; e < hi; e += 256) a[e] = b[e];
}

int main(int argc, char **argv) {
    const size_t n2 = (size_t)1 << 26;                 // 2^26 double2 = 1 GiB per array
    double2 *a, *b, *c; double *out;
    CK(hipMalloc(&a, n2 * 16)); CK(hipMalloc(&b, n2 * 16)); CK(hipMalloc(&c, n2 * 16)); CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 0, n2 * 16)); CK(hipMemset(b, 0, n2 * 16)); CK(hipMemset(c, 0, n2 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 10;
    auto timeit = [&](const char *name, double bytes, int grid, auto launch) {
        for (int i = 0; i < 2; ++i) launch(grid);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch(grid);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-14s grid %6d : %7.1f us  %6.2f TB/s\n", name, grid, ms / reps * 1e3, bytes * reps / (ms * 1e-3) / 1e12);
    };
    const double GB = (double)n2 * 16;
    for (int grid : {256, 512, 1024, 2048, 4096, 8192, 16384, 65536}) {
        timeit("read", GB, grid, [&](int g) { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, out, n2); });
        timeit("write", GB, grid, [&](int g) { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, a, n2); });
        timeit("write_nt", GB, grid, [&](int g) { hipLaunchKernelGGL(k_write_nt, dim3(g), dim3(256), 0, 0, a, n2); });
        timeit("copy", 2 * GB, grid, [&](int g) { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n2); });
        timeit("copy_u4", 2 * GB, grid, [&](int g) { hipLaunchKernelGGL(k_copy_u<4>, dim3(g), dim3(256), 0, 0, a, b, n2); });
        timeit("copy_blk", 2 * GB, grid, [&](int g) { hipLaunchKernelGGL(k_copy_blk, dim3(g), dim3(256), 0, 0, a, b, n2); });
        timeit("triad", 3 * GB, grid, [&](int g) { hipLaunchKernelGGL(k_triad, dim3(g), dim3(256), 0, 0, a, b, c, 1.5, n2); });
    }
    return 0;
}
