// HIP kernels of libwaehip.so -- written for gfx950 (MI355X, CDNA4; wave64) only.
//
// Data layout.  Every multi-vector is "interleaved": X[row][b], b = 0..nb-1 complex doubles contiguous per
// row, so that the gather X[col[p]][0..nb) of one matrix nonzero is ONE contiguous nb*16-byte segment
// (nb = 8 -> one 128-B line) and all vector kernels are plain coalesced streams.
// Operators are sums of "planes" (term matrices) grouped by shared sparsity pattern: a group stores
// rowptr/col once and its planes' values interleaved per nonzero ([nnz][nplanes], real planes as 8-B doubles),
// so the fused multi-term SpMV reads each index once and all term values of that nonzero in one load.
#include <mutex>
#include <type_traits>
#include "wae_internal.h"

#include <cstdlib>

// ---------------------------------------------------------------------------------------------------
// complex helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return cplx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cplx cconj(cplx a) { return cplx{a.x, -a.y}; }
__device__ __forceinline__ void cfma(cplx &acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x); acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y); acc.y = fma(a.y, b.x, acc.y);
}
// streaming (non-temporal) read of a vector entry: the Krylov basis is read once per kernel and is far larger than any cache
// (1.1 % of a 1M-DoF pass, paired runs; the same hint on w and on the store of the update: nothing measurable)
typedef double dbl2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cplx stream_load(const cplx *p) {
#ifndef WAE_NO_NT_GS
    const dbl2v v = __builtin_nontemporal_load((const dbl2v *)p);
    return cplx{v.x, v.y};
#else
    return *p;
#endif
}
__device__ __forceinline__ cplx cdiv(cplx a, cplx b) {
    double s = 1.0 / (b.x * b.x + b.y * b.y);
    return cplx{(a.x * b.x + a.y * b.y) * s, (a.y * b.x - a.x * b.y) * s};
}

// Long rows (OpDev::long_*, TileDev::ls_*): acc[row][b] = sum over the entries of the row of pc[sys(b)][slot] * val * X[col][b].  A row of
// the transposed flame term holds one entry per flame node (4 875 at 1M DoF, 48 such rows): as ONE workgroup's loop -- 150 dependent
// index -> operand round trips per lane -- that was 68-74 us per product:
// a sixth of an adjoint Krylov step of the Newton-type solvers, whose narrow batches leave the rest of the chip idle meanwhile.  The
// entries of a row are therefore split over WAE_LONG_SPLIT workgroups per 8-column chunk (32 lanes stride over a piece, 8 lanes across
// the columns; fixed-order LDS reduction) that write partial sums, and long_reduce_kernel adds the partials in a fixed order.
__device__ __forceinline__ void long_row_piece(const int *__restrict__ ptr, const int *__restrict__ col, const int *__restrict__ slot,
                                               const cplx *__restrict__ val, int npl, int conj, const cplx *__restrict__ pc, int cps,
                                               const cplx *__restrict__ X, int nb, const unsigned char *__restrict__ cmask, cplx *__restrict__ part) {
    __shared__ cplx red[256];
    const int li = blockIdx.x / WAE_LONG_SPLIT, k = blockIdx.x - li * WAE_LONG_SPLIT, ch = blockIdx.y;
    if (cmask && !cmask[ch]) return;
    const int tid = threadIdx.x, c = tid & 7, seg = tid >> 3;
    const int b = ch * 8 + c;
    const int bb = b < nb ? b : nb - 1;
    const cplx *mypc = pc + (size_t)(bb / cps) * npl;
    const double sg = conj ? -1.0 : 1.0;
    const int beg = ptr[li], end = ptr[li + 1];
    const int piece = ((end - beg + WAE_LONG_SPLIT - 1) / WAE_LONG_SPLIT + 31) & ~31;
    const int p0 = beg + k * piece, p1 = min(end, p0 + piece);
    cplx acc = {0.0, 0.0};
    int p = p0 + seg;
    for (; p + 96 < p1; p += 128) {                      // four entries in flight per lane
        int cc[4], sl[4];
        cplx a[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { cc[u] = col[p + 32 * u]; sl[u] = slot[p + 32 * u]; a[u] = val[p + 32 * u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = X[(size_t)cc[u] * nb + bb];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u].y *= sg; cfma(acc, cmul(mypc[sl[u]], a[u]), x[u]); }
    }
    for (; p < p1; p += 32) {
        cplx a = val[p];
        a.y *= sg;
        cfma(acc, cmul(mypc[slot[p]], a), X[(size_t)col[p] * nb + bb]);
    }
    red[tid] = acc;
    __syncthreads();
    for (int off = 128; off >= 8; off >>= 1) {
        if (tid < off) { red[tid].x += red[tid + off].x; red[tid].y += red[tid + off].y; }
        __syncthreads();
    }
    if (tid < 8 && b < nb) part[((size_t)li * WAE_LONG_SPLIT + k) * nb + b] = red[tid];
}
__global__ __launch_bounds__(256) void spmv_long_kernel(OpDev op, const cplx *__restrict__ pc, int cps, const cplx *__restrict__ X, int nb,
                                                        const unsigned char *__restrict__ cmask) {
    long_row_piece(op.long_ptr, op.long_col, op.long_slot, op.long_val, op.nplanes_total, op.long_conj, pc, cps, X, nb, cmask, op.long_part);
}
// acc[map ? map[i] : i][b] = sum_k part[i][k][b], k ascending (one thread per (row, column))
__global__ __launch_bounds__(256) void long_reduce_kernel(const cplx *__restrict__ part, int nrows, int nb, const int *__restrict__ map,
                                                          cplx *__restrict__ acc, const unsigned char *__restrict__ cmask) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nrows * nb) return;
    const int i = e / nb, b = e - i * nb;
    if (cmask && !cmask[b >> 3]) return;
    cplx s = {0.0, 0.0};
    for (int k = 0; k < WAE_LONG_SPLIT; ++k) { const cplx t = part[((size_t)i * WAE_LONG_SPLIT + k) * nb + b]; s.x += t.x; s.y += t.y; }
    acc[(size_t)(map ? map[i] : i) * nb + b] = s;
}
static void launch_long_rows(const OpDev &op, const cplx *pc, int cps, const cplx *X, int nb, const unsigned char *cmask, hipStream_t st) {
    hipLaunchKernelGGL(spmv_long_kernel, dim3((unsigned)op.nlong * WAE_LONG_SPLIT, (unsigned)((nb + 7) / 8)), dim3(256), 0, st, op, pc, cps, X, nb, cmask);
    HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(long_reduce_kernel, dim3((unsigned)((op.nlong * nb + 255) / 256)), dim3(256), 0, st, op.long_part, op.nlong, nb,
                       (const int *)nullptr, op.long_acc, cmask);
    HIP_CHECK(hipGetLastError());
}
// position of `row` in the sorted long-row list, or -1
__device__ __forceinline__ int long_row_index(const OpDev &op, int64_t row) {
    int lo = 0, hi = op.nlong - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int v = op.long_rows[mid];
        if (v == row) return mid;
        if (v < row) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------------
// fused multi-term CSR SpMV / SpMM:   Y = f( sum_q pc[sys(b)][q] * plane_q * X )   for nb columns
//
// A "team" of C*S lanes owns one matrix row: C lanes across batch columns (adjacent lanes -> adjacent X
// columns: the gather of one nonzero is a single contiguous C*16-byte access), S lanes across the row's
// nonzeros (partial sums combined with wave shuffles).  64/(C*S) rows per wavefront, 256-thread blocks.
// Per-column plane coefficients are staged once per block in LDS.
// ---------------------------------------------------------------------------------------------------
template <int C, int S>
__global__ __launch_bounds__(256) void spmv_kernel(OpDev op, const cplx *__restrict__ pc, int cps,
                                                   const cplx *__restrict__ X, cplx *Y,
                                                   const cplx *B, double jac_w, int nb, int mode,
                                                   const unsigned char *__restrict__ cmask) {
    if (C == 8 && cmask && !cmask[blockIdx.y]) return;      // whole 8-column chunk converged (uniform per workgroup)
    constexpr int TEAM = C * S;
    constexpr int TPB = 256 / TEAM;
    extern __shared__ cplx spc[];   // [C][nplanes_total]
    const int tid = threadIdx.x;
    const int npl = op.nplanes_total;
    for (int i = tid; i < C * npl; i += 256) {
        int cc = i / npl, q = i - cc * npl;
        int bb = blockIdx.y * C + cc;
        spc[i] = (bb < nb) ? pc[(size_t)(bb / cps) * npl + q] : cplx{0.0, 0.0};
    }
    __syncthreads();
    const int team = tid / TEAM;
    const int lt = tid - team * TEAM;
    const int c = lt % C;
    const int s = lt / C;
    // XCD-aware row-block order: workgroups are dealt round-robin over the 8 XCDs (gridDim.x is a multiple of 8),
    // so give XCD k the k-th contiguous eighth of the rows: its L2 then holds one slab of X instead of all of it.
    const unsigned per_xcd = gridDim.x >> 3;
    const int64_t rb = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const int64_t row = rb * TPB + team;
    if (row >= op.n) return;        // whole teams leave together (shuffles below stay inside a team)
    const int b = blockIdx.y * C + c;
    const bool active = b < nb;
    const int bb = active ? b : nb - 1;
    const cplx *mypc = spc + c * npl;
    cplx acc = {0.0, 0.0};
#pragma unroll 1
    for (int g = 0; g < op.ngroups; ++g) {
        const GroupDev G = op.g[g];
        const int p0 = G.rowptr[row], p1 = G.rowptr[row + 1];
        const int np = G.nplanes;
        const cplx *gpc = mypc + G.plane0;
        if (G.is_real) {
            const double *__restrict__ v = (const double *)G.vals;
            if (np == 2) {          // the hot case: mass + stiffness share one pattern (16 B per nonzero)
                const double2 *__restrict__ v2 = (const double2 *)G.vals;
                const cplx c0 = gpc[0], c1 = gpc[1];
                // 4 nonzeros per trip: issue all index/value loads, then all X gathers, then the FMAs, so that four
                // dependent (col -> X) chains are in flight per lane instead of one
                constexpr int U = 4;
                for (int p = p0 + s; p < p1; p += U * S) {
                    int j[U];
                    double2 a[U];
                    cplx x[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int pp = p + u * S;
                        const int pc_ = pp < p1 ? pp : p1 - 1;
                        j[u] = G.col[pc_];
                        a[u] = v2[pc_];
                        if (pp >= p1) a[u] = double2{0.0, 0.0};
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) x[u] = X[(size_t)j[u] * nb + bb];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        cplx m = {fma(c0.x, a[u].x, c1.x * a[u].y), fma(c0.y, a[u].x, c1.y * a[u].y)};
                        cfma(acc, m, x[u]);
                    }
                }
            } else {
                for (int p = p0 + s; p < p1; p += S) {
                    const int j = G.col[p];
                    cplx m = {0.0, 0.0};
                    for (int q = 0; q < np; ++q) {
                        const double a = v[(size_t)p * np + q];
                        m.x = fma(gpc[q].x, a, m.x);
                        m.y = fma(gpc[q].y, a, m.y);
                    }
                    const cplx x = X[(size_t)j * nb + bb];
                    cfma(acc, m, x);
                }
            }
        } else {
            const cplx *__restrict__ v = (const cplx *)G.vals;
            const double sg = G.conj_vals ? -1.0 : 1.0;
            for (int p = p0 + s; p < p1; p += S) {
                const int j = G.col[p];
                cplx m = {0.0, 0.0};
                for (int q = 0; q < np; ++q) {
                    cplx a = v[(size_t)p * np + q];
                    a.y *= sg;
                    cfma(m, gpc[q], a);
                }
                const cplx x = X[(size_t)j * nb + bb];
                cfma(acc, m, x);
            }
        }
    }
#pragma unroll
    for (int off = C; off < TEAM; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off);
        acc.y += __shfl_xor(acc.y, off);
    }
    if (s != 0 || !active) return;
    if (op.nlong) {                                          // this row's long part was summed by spmv_long_kernel
        const int li = long_row_index(op, row);
        if (li >= 0) { const cplx t = op.long_acc[(size_t)li * nb + b]; acc.x += t.x; acc.y += t.y; }
    }
    const size_t e = (size_t)row * nb + b;
    cplx out;
    if (mode == MODE_AX) {
        out = acc;
    } else if (mode == MODE_RES) {
        const cplx bv = B[e];
        out = cplx{bv.x - acc.x, bv.y - acc.y};
    } else if (mode == MODE_ADD) {
        const cplx bv = B[e];
        out = cplx{bv.x + acc.x, bv.y + acc.y};
    } else {   // MODE_JAC / MODE_AX_DS / MODE_RES_DS need the diagonal of this column's operator
        cplx dg = {0.0, 0.0};
        const double dsg = op.conj_diag ? -1.0 : 1.0;
        for (int q = 0; q < npl; ++q) { cplx dq = op.diag[(size_t)row * npl + q]; dq.y *= dsg; cfma(dg, mypc[q], dq); }
        if (mode == MODE_AX_DS) {
            out = cdiv(acc, dg);
        } else if (mode == MODE_RES_DS) {
            const cplx bv = B[e];
            out = cdiv(cplx{bv.x - acc.x, bv.y - acc.y}, dg);
        } else {
            const cplx bv = B[e], xv = X[e];
            cplx r = cdiv(cplx{bv.x - acc.x, bv.y - acc.y}, dg);
            out = cplx{xv.x + jac_w * r.x, xv.y + jac_w * r.y};
        }
    }
    Y[e] = out;
}

// ---------------------------------------------------------------------------------------------------
// LDS-staged variant for batch widths >= 8 (C = 8 column lanes per row, one row per team).
// The L1 -> VGPR return path (64 B/clk/CU) is what limits the plain kernel at C = 8: every column lane of a team
// receives its own copy of the same index and values.  Here the workgroup streams the (col, values) of its rows
// ONCE, fully coalesced, into LDS (chunks of up to LDS_NNZ nonzeros), and the teams then read them back as LDS
// broadcasts (conflict-free: the 8 lanes of a team hit one address); the only global gathers left are the X rows.
// ---------------------------------------------------------------------------------------------------
constexpr int LDS_WORDS = 2048;   // 8-byte value words staged per chunk (16 KB)
constexpr int LDS_NNZ = 1024;     // nonzeros staged per chunk (4 KB of indices)

// One workgroup = 32 rows x (8*NCH) batch columns: the staged matrix tile is reused for NCH column chunks, so the
// matrix streams through the workgroup once per 8*NCH columns (once per launch at the default batch width 64) and
// each gathered X row is consumed as NCH consecutive 128-B segments of one 1-KB interleaved row.
// __launch_bounds__(256, 4): four workgroups per CU.  An ablation showed that the launch time is the sum of the
// per-workgroup phases (stage -> barrier -> gather/FMA loop -> store) times workgroups-per-CU over resident workgroups -- a
// latency chain, not a bandwidth limit -- so residency matters: at 130 VGPRs the compiler's default gave 3 waves per SIMD,
// the bound makes it fit 126 VGPRs without spilling (4 waves: -10 %; forcing 5 spills or halves the unrolling: slower).
template <int NCH>
__global__ __launch_bounds__(256, 4) void spmv_lds_kernel(OpDev op, const cplx *__restrict__ pc, int cps,
                                                       const cplx *__restrict__ X, cplx *Y, const cplx *B, double jac_w,
                                                       int nb, int mode, const unsigned char *__restrict__ cmask) {
    constexpr int C = 8;
    constexpr int TPB = 256 / C;            // rows per workgroup
    constexpr int CW = C * NCH;             // columns per workgroup
    if (cmask) {                            // skip the workgroup if all of its column chunks have converged (uniform)
        bool any = false;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = blockIdx.y * NCH + k;
            if (ch * C < nb && cmask[ch]) any = true;
        }
        if (!any) return;
    }
    extern __shared__ cplx spc[];           // [CW][nplanes_total]
    __shared__ __attribute__((aligned(16))) double lds_w[LDS_WORDS];
    __shared__ int lds_c[LDS_NNZ];
    const int tid = threadIdx.x;
    const int npl = op.nplanes_total;
    const int col0 = blockIdx.y * CW;
    for (int i = tid; i < CW * npl; i += 256) {
        int cc = i / npl, q = i - cc * npl;
        int bb = col0 + cc;
        spc[i] = (bb < nb) ? pc[(size_t)(bb / cps) * npl + q] : cplx{0.0, 0.0};
    }
    const int team = tid / C;
    const int c = tid - team * C;
    const unsigned per_xcd = gridDim.x >> 3;
    const int64_t rb = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const int64_t row0 = rb * TPB;
    const int64_t row = row0 + team;
    const bool valid = row < op.n;
    int bcol[NCH];                          // this lane's column in every chunk (clamped for loads)
    bool act[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int b = col0 + k * C + c;
        act[k] = b < nb && (!cmask || cmask[blockIdx.y * NCH + k]);
        bcol[k] = b < nb ? b : nb - 1;
    }
    cplx acc[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) acc[k] = cplx{0.0, 0.0};
    const int64_t rlast = (row0 + TPB < op.n) ? row0 + TPB : op.n;     // first row after this block
    __syncthreads();
    if (row0 < op.n) {
#pragma unroll 1
        for (int g = 0; g < op.ngroups; ++g) {
            const GroupDev G = op.g[g];
            const int np = G.nplanes;
            const int wpn = G.is_real ? np : 2 * np;                  // 8-byte words per nonzero
            int chunk = LDS_WORDS / wpn;
            if (chunk > LDS_NNZ) chunk = LDS_NNZ;
            const int blo = G.rowptr[row0], bhi = G.rowptr[rlast];
            const int p0 = valid ? G.rowptr[row] : 0, p1 = valid ? G.rowptr[row + 1] : 0;
            const double *__restrict__ gw = (const double *)G.vals;
            const double sg = G.conj_vals ? -1.0 : 1.0;
            for (int lo = blo; lo < bhi; lo += chunk) {
                const int hi = (lo + chunk < bhi) ? lo + chunk : bhi;
                const int cnt = hi - lo;
                __syncthreads();                                      // previous chunk fully consumed
                for (int i = tid; i < cnt; i += 256) lds_c[i] = G.col[lo + i];
                const int nw = cnt * wpn;
                const double *src = gw + (size_t)lo * wpn;
                for (int i = tid; i < nw; i += 256) lds_w[i] = src[i];
                __syncthreads();
                const int a0 = p0 > lo ? p0 : lo, a1 = p1 < hi ? p1 : hi;
                if (G.is_real && np == 2) {
                    const double2 *lv = (const double2 *)lds_w;
                    constexpr int U = (NCH >= 4) ? 2 : 4;             // U*NCH gathers in flight per lane
                    for (int p = a0; p < a1; p += U) {
                        int j[U];
                        double2 a[U];
                        cplx x[U][NCH];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int pp = (p + u < a1) ? p + u : a1 - 1;
                            j[u] = lds_c[pp - lo];
                            a[u] = lv[pp - lo];
                            if (p + u >= a1) a[u] = double2{0.0, 0.0};
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u)
#pragma unroll
                            for (int k = 0; k < NCH; ++k) x[u][k] = act[k] ? X[(size_t)j[u] * nb + bcol[k]] : cplx{0.0, 0.0};
#pragma unroll
                        for (int u = 0; u < U; ++u)
#pragma unroll
                            for (int k = 0; k < NCH; ++k) {
                                const cplx *gpc = spc + (k * C + c) * npl + G.plane0;
                                const cplx c0 = gpc[0], c1 = gpc[1];
                                cplx m = {fma(c0.x, a[u].x, c1.x * a[u].y), fma(c0.y, a[u].x, c1.y * a[u].y)};
                                cfma(acc[k], m, x[u][k]);
                            }
                    }
                } else {
                    for (int p = a0; p < a1; ++p) {
                        const int j = lds_c[p - lo];
#pragma unroll
                        for (int k = 0; k < NCH; ++k) {
                            const cplx *gpc = spc + (k * C + c) * npl + G.plane0;
                            cplx m = {0.0, 0.0};
                            if (G.is_real) {
                                for (int q = 0; q < np; ++q) {
                                    const double a = lds_w[(p - lo) * np + q];
                                    m.x = fma(gpc[q].x, a, m.x);
                                    m.y = fma(gpc[q].y, a, m.y);
                                }
                            } else {
                                const cplx *lv = (const cplx *)lds_w;
                                for (int q = 0; q < np; ++q) {
                                    cplx a = lv[(p - lo) * np + q];
                                    a.y *= sg;
                                    cfma(m, gpc[q], a);
                                }
                            }
                            if (act[k]) cfma(acc[k], m, X[(size_t)j * nb + bcol[k]]);
                        }
                    }
                }
            }
        }
    }
    if (!valid) return;
    const int li_long = op.nlong ? long_row_index(op, row) : -1;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (!act[k]) continue;
        const int b = col0 + k * C + c;
        if (li_long >= 0) { const cplx t = op.long_acc[(size_t)li_long * nb + b]; acc[k].x += t.x; acc[k].y += t.y; }
        const cplx *mypc = spc + (k * C + c) * npl;
        const size_t e = (size_t)row * nb + b;
        cplx out;
        if (mode == MODE_AX) {
            out = acc[k];
        } else if (mode == MODE_RES) {
            const cplx bv = B[e];
            out = cplx{bv.x - acc[k].x, bv.y - acc[k].y};
        } else if (mode == MODE_ADD) {
            const cplx bv = B[e];
            out = cplx{bv.x + acc[k].x, bv.y + acc[k].y};
        } else {   // MODE_JAC / MODE_AX_DS / MODE_RES_DS need the diagonal of this column's operator
            cplx dg = {0.0, 0.0};
            const double dsg = op.conj_diag ? -1.0 : 1.0;
            for (int q = 0; q < npl; ++q) { cplx dq = op.diag[(size_t)row * npl + q]; dq.y *= dsg; cfma(dg, mypc[q], dq); }
            if (mode == MODE_AX_J0) {
                out = acc[k];
                const cplx r = cdiv(acc[k], dg);
                const_cast<cplx *>(B)[e] = cplx{jac_w * r.x, jac_w * r.y};
            } else if (mode == MODE_AX_DS) {
                out = cdiv(acc[k], dg);
            } else if (mode == MODE_RES_DS) {
                const cplx bv = B[e];
                out = cdiv(cplx{bv.x - acc[k].x, bv.y - acc[k].y}, dg);
            } else {
                const cplx bv = B[e], xv = X[e];
                cplx r = cdiv(cplx{bv.x - acc[k].x, bv.y - acc[k].y}, dg);
                out = cplx{xv.x + jac_w * r.x, xv.y + jac_w * r.y};
            }
        }
        Y[e] = out;
    }
}

// Side rows of the tile path (TileDev): side_acc[i][b] = sum over the entries of side row i of pc[sys(b)][slot] * val * X[col][b].
// Thread = (side row, column): 8 lanes read one 128-B segment of an X row.
__global__ __launch_bounds__(256) void spmv_side_kernel(TileDev td, int npl, int conj, const cplx *__restrict__ pc, int cps,
                                                        const cplx *__restrict__ X, int nb, const unsigned char *__restrict__ cmask) {
    const int ch = blockIdx.y;
    if (cmask && !cmask[ch]) return;
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3), b = ch * 8 + (threadIdx.x & 7);
    if (i >= td.nside || b >= nb) return;
    const cplx *mypc = pc + (size_t)(b / cps) * npl;
    const double sg = conj ? -1.0 : 1.0;
    cplx acc = {0.0, 0.0};
    const int beg = td.side_ptr[i], end = td.side_ptr[i + 1];
    for (int p0 = beg; p0 < end; p0 += 8) {                  // eight entries at a time: three memory latencies per batch (entry,
        int c[8], sl[8];                                     // operand, -) instead of two per entry
        cplx a[8], x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + u < end ? p0 + u : end - 1;
            c[u] = td.side_col[p]; sl[u] = td.side_slot[p]; a[u] = td.side_val[p];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = X[(size_t)c[u] * nb + b];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (p0 + u < end) { a[u].y *= sg; cfma(acc, cmul(mypc[sl[u]], a[u]), x[u]); }
    }
    td.side_acc[(size_t)i * nb + b] = acc;
}

// Long side rows (TileDev::nlong_side; the transposed orientation of a flame term): side_acc[ls_side[li]][b] = the row's sum, in pieces
// like spmv_long_kernel (long_reduce_kernel then writes the sums over the zeros spmv_side_kernel left there: empty CSR rows).
__global__ __launch_bounds__(256) void spmv_side_long_kernel(TileDev td, int npl, int conj, const cplx *__restrict__ pc, int cps,
                                                             const cplx *__restrict__ X, int nb, const unsigned char *__restrict__ cmask) {
    long_row_piece(td.ls_ptr, td.ls_col, td.ls_slot, td.ls_val, npl, conj, pc, cps, X, nb, cmask, td.ls_part);
}

// ---------------------------------------------------------------------------------------------------
// Tile kernel for batch widths >= 8 on an operator whose rows have been renumbered into tiles (tiles.h): the fine level (LPR = 2
// lanes per row), the first coarse level and the fine-to-coarse restriction (LPR = 4).  DESIGN.md 4b tells how it got here.
//
// PERSISTENT workgroups of 8 wavefronts, one per CU; each draws tiles (<= 512 / LPR consecutive rows forming a compact brick of
// the mesh graph) and walks a tile's chunks of 8 batch columns.
//  * The matrix slice of a tile's rows is loaded ONCE into registers and serves every chunk: wavefront w owns rows
//    (64 / LPR) w .. (64 / LPR)(w + 1) - 1; the LPR lanes of a row split its entries (lane LPR i + h: entries h, h + LPR, ...;
//    8 / 12 register-resident entries per lane cover 16 / 48 per row, longer rows stream the rest one entry ahead).
//  * The tile's window -- the distinct X rows its rows touch (~2 x 256 on the fine level), 128 B each per chunk -- is gathered
//    into LDS by LDS-DMA (global_load_lds_dwordx4: the per-lane source address makes it a row gather, no VGPR round trip).
//    NBUF window buffers form a ring: the gather for the chunk NBUF - 1 ahead is issued piece by piece between the entries of
//    the current chunk.
//  * Compute: per entry one coefficient product and 8 x (ds_read_b128 + complex FMA), software-pipelined by hand (operands of
//    entry u+1 requested before the FMAs of entry u).  The lanes read the columns in a rotated order so that the 16 lanes a
//    ds_read_b128 serves per LDS cycle spread over all bank groups; partner lanes share a group only if their window slots
//    have equal parity, which the entry order avoids (tiles.cpp).
//  * The partial sums of a row meet through DPP quad permutations -- no LDS staging, no second barrier -- and each lane writes
//    8 / LPR of the row's eight results (the lanes of a row complete one 128-B segment): ONE barrier per chunk, and the stores
//    of chunk c drain under the compute of chunk c+1.
//  * The rows of the other pattern groups come from spmv_side_kernel (above).
// UNI: the 8 columns of a chunk belong to one system (columns per system a multiple of 8): one coefficient set per chunk.
// ---------------------------------------------------------------------------------------------------
#ifdef WAE_TILE_STAMPS
__device__ unsigned long long wae_tile_stamps[8 * 64 + 8]; // diagnostic build only: s_memtime at the phase boundaries of one workgroup
#ifndef TILE_STAMP_WG
#define TILE_STAMP_WG 101
#endif
__device__ unsigned long long wae_tile_wglog[1024 * 4];    // per workgroup: start, end (100-MHz clock), chunks done, HW_ID | XCC_ID << 32
#ifndef TILE_STAMP_FIRST
#define TILE_STAMP_FIRST 21        /* chunks 21..28 of that workgroup: with 8 chunks per tile, the switch from its 3rd to its 4th tile is among them */
#endif
#define TILE_STAMP(k) do { if (blockIdx.x == TILE_STAMP_WG && tid == 0 && nstamp >= TILE_STAMP_FIRST && nstamp < TILE_STAMP_FIRST + 8) { \
                               wae_tile_stamps[(nstamp - TILE_STAMP_FIRST) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
                               if ((k) == 0) wae_tile_stamps[(nstamp - TILE_STAMP_FIRST) * 8 + 7] = __builtin_amdgcn_s_memrealtime(); } } while (0)   /* [7]: the 100-MHz clock */
#else
#define TILE_STAMP(k) do { } while (0)
#endif
template <int CTRL> __device__ __forceinline__ double lane_quad(double v) {     // DPP quad_perm CTRL: a value from another lane of the quad
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// DPP move of a double under a bank mask (a bank = four lanes of a 16-lane row): enabled lanes take `src` from the lane CTRL names,
// the others keep `old`
template <int CTRL, int BANKS> __device__ __forceinline__ double lane_dpp_masked(double old, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANKS, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANKS, false);
    return __hiloint2double(hi, lo);
}
// LPR = lanes per row: 2 (rows of up to 16 register-resident entries, 256-row tiles: the fine level) or 4 (up to 48, 128-row
// tiles: the first coarse level, whose windows allow ~64 rows per tile anyway).
// NWV = wavefronts per workgroup: 8, or 16 ("QUAD", round 3; fine level, one system per chunk): the tile keeps its 256 rows, its
// windows and its storage (two entry halves per row), but every row is worked on by FOUR lanes -- lane 4 i + 2 c + h takes entry half
// h and column half c (four of the chunk's eight columns) -- so that a lane carries 16 accumulator and 16 operand registers instead
// of 32 and 64, the kernel fits 128 VGPRs, and four wavefronts per SIMD instead of two work on the chain barrier -> entries ->
// window landed -> stores (that chain, not a bandwidth, is what the 8-wavefront form waits for: two wavefronts per SIMD sit in
// s_waitcnt half of their time).  Per wavefront and chunk half the LDS reads and FMAs; the two c lanes of a half read the same
// matrix entries (one coalesced request).
template <bool UNI, int LPR, int NBUF, int NWV = 8, int KRT = (LPR == 2 ? 8 : 12)>
__global__ __launch_bounds__(64 * NWV, NWV == 8 ? 1 : 4) void spmv_tile_kernel(OpDev op, TileDev td, const cplx *__restrict__ pc, int cps,
                                                          const cplx *__restrict__ X, cplx *Y, const cplx *B, double jac_w,
                                                          int nb, int mode, const unsigned char *__restrict__ cmask, int spc_all, int csplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];
    const int tid = threadIdx.x;
#ifdef WAE_TILE_STAMPS
    if (blockIdx.x == TILE_STAMP_WG && tid == 0) wae_tile_stamps[64] = __builtin_amdgcn_s_memtime();      // kernel entry
    if (tid == 0 && blockIdx.x < 1024) {
        wae_tile_wglog[blockIdx.x * 4] = __builtin_amdgcn_s_memrealtime();
        wae_tile_wglog[blockIdx.x * 4 + 3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
    }
#endif
    const int nch_all = (nb + 7) >> 3;
    const int npl = op.nplanes_total;
    const int wslots = (td.wmax + 7) & ~7;
    cplx *const smem = (cplx *)tile_smem;                   // NBUF window buffers [window slot][8 columns] (offsets, not a pointer
    cplx *spc0 = smem + (size_t)wslots * 8 * NBUF;          // table: the accesses must stay provably LDS), then the coefficients:
                                                            // [8][npl] of the current chunk, or (spc_all) [nb][npl] staged once
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: branches on it are uniform)
    constexpr int RPW = 64 / LPR;                            // rows per wavefront
    constexpr int NOUT = 8 / LPR;                            // results per lane and chunk
    const int sub = lane & (LPR - 1);                        // which share of its row's entries (and of its results) this lane takes
    // Column order of the LDS reads.  LPR = 2: rotated by the row number, the same for the two lanes of a row (their partial
    // sums line up position by position; they share a bank group only if their window slots have equal parity, which the
    // entry order avoids).  LPR = 4: lane q of a row starts 2 q further on, rows alternate between the even and the odd
    // columns: position s of lane q is column s + rot, and the four partial sums of a column sit at positions that differ by 2
    // from lane to lane -- a reduce-scatter through three quad rotations leaves each lane with the two columns it stores.
    const int rot = LPR == 2 ? (lane >> 1) & 7 : (((lane >> 2) & 1) + 2 * sub) & 7;
    const int rot_out = LPR == 2 ? (rot + 4 * sub) & 7 : rot;    // column of this lane's first result
    // Work list.  The workgroups are PERSISTENT: the grid has one per CU, dealt round-robin over the 8 XCDs by the hardware.
    // XCD k owns the k-th contiguous eighth of the tiles (its "share"); its workgroups take the first positions of the share
    // statically and then draw the next position from a counter (td.counters[k]), so that they walk the share side by side --
    // neighbouring tiles (shared halo rows) are in flight on one L2 at about the same time -- at their own pace: the time per
    // chunk differs between CUs (measured, identical work).  A workgroup whose share is used up draws from the other shares.
    // The LAST tiles of a share -- as many as the XCD has workgroups -- are handed out in `csplit` parts of their chunks each, so
    // that the launch ends within a fraction of a tile's time on every CU (200k unknowns: three tiles per CU on average, four
    // on some); with fewer tiles than CUs that is all of them.  A work item = (share, position) -> (tile, part).  The last
    // workgroup to leave zeroes the counters for the next launch (launches that use one operator are ordered by its stream).
    const int tpx = (td.ntiles + 7) >> 3;
    const int xcd = (int)(blockIdx.x & 7u), gpx = (int)(gridDim.x >> 3);
    auto next_active = [&](int c, int end) { while (c < end && cmask && !cmask[c]) ++c; return c; };
    auto share_tiles = [&](int x) { const int left = td.ntiles - x * tpx; return left < tpx ? (left > 0 ? left : 0) : tpx; };
    auto share_split = [&](int x) { const int n = share_tiles(x); return csplit > 1 ? (n < gpx ? n : gpx) : 0; };   // tiles handed out in parts
    auto share_size = [&](int x) { return share_tiles(x) + share_split(x) * (csplit - 1); };
    auto decode = [&](int item, int &tile_, int &ch_, int &end_) {    // item = share << 24 | position;  -> has an active chunk
        const int x = item >> 24, p = item & 0xFFFFFF;
        const int nfull = share_tiles(x) - share_split(x);
        int part = 0, parts = 1;
        if (p < nfull) tile_ = x * tpx + p;
        else { const int q = p - nfull; tile_ = x * tpx + nfull + q / csplit; part = q - (q / csplit) * csplit; parts = csplit; }
        end_ = (part + 1) * nch_all / parts;
        ch_ = next_active(part * nch_all / parts, end_);
        return ch_ < end_;
    };
    auto draw = [&](int pend) {                              // (one thread) next work item with work, or -1; pend: position already
        int t_, c_, e_;                                      // drawn from the own share, or -1
        if (pend >= 0 && pend < share_size(xcd) && decode(xcd << 24 | pend, t_, c_, e_)) return xcd << 24 | pend;
        for (int a = 0; a < 8; ++a) {
            const int x2 = (xcd + a) & 7, lim = share_size(x2);
            while (gpx < lim) {
                const int p = gpx + (int)atomicAdd(td.counters + x2, 1u);
                if (p >= lim) break;
                if (decode(x2 << 24 | p, t_, c_, e_)) return x2 << 24 | p;
            }
        }
        return -1;
    };
    int *const vt_slot = (int *)(spc0 + (size_t)(spc_all ? nch_all * 8 : 8) * npl);   // LDS word: the next virtual tile of this workgroup
    auto leave = [&]() {                                     // (every workgroup, once)
        if (tid == 0 && atomicAdd(td.counters + 8, 1u) == gridDim.x - 1)
            for (int i = 0; i < 9; ++i) atomicExch(td.counters + i, 0u);
    };
    int tile, ch, ch_end;
    {
        const int p0 = (int)(blockIdx.x >> 3);
        bool ok = p0 < share_size(xcd) && decode(xcd << 24 | p0, tile, ch, ch_end);
        if (!ok) {                                           // (no static tile, or none of its chunks active)
            if (tid == 0) *vt_slot = draw(-1);
            __syncthreads();
            const int vt = *vt_slot;
            __syncthreads();
            ok = vt >= 0 && decode(vt, tile, ch, ch_end);
        }
        if (!ok) { leave(); return; }
    }
    constexpr int KR = KRT;                                  // register-resident entries per lane (16 / 48 per row; 16 with NWV = 16)
    constexpr int WSTEP = 8 * NWV;                           // window rows per workgroup step (8 per wave instruction)
    constexpr int NTHR = 64 * NWV;
    constexpr int NW = NBUF > 2 ? 7 : 640 / WSTEP;           // windows up to 448 / 640 rows in one sweep
    constexpr bool QUAD = NWV == 16;                         // four lanes per row: (entry half h) x (column half c); LPR = 4 (16 rows per wavefront)
    static_assert(!QUAD || (LPR == 4 && UNI && NBUF == 2 && KRT == 8), "QUAD: LPR = 4, one system per chunk, two buffers, 8 entries per lane");
    const int qh = lane & 1, qc = (lane >> 1) & 1, qrot = (lane >> 2) & 3;     // (QUAD) this lane's halves, and the rotation of its row
    // column of result j of this lane: QUAD keeps positions 2 h + j of its column half (position s <-> column 4 c + (s + qrot) mod 4)
    // COAL (round 4; one system per chunk, two lanes per row, two buffers -- the form of every wide solve): the epilogue touches memory
    // ROW-CONTIGUOUSLY.  In the older form each lane wrote its four results of ONE row, 16 bytes at a time: a wave instruction was 64
    // separate 16-byte requests (counters: 63.7 M of the 84.5 M L2 requests of a product at 1M unknowns were such partial writes), and the
    // right-hand sides of the fused forms were read the same way.  Here the two lanes of a row keep the even / the odd COLUMNS (not
    // the lower / upper positions), the four rows of an 8-lane group are transposed through DPP (quad permute for the neighbouring pair,
    // masked row shifts by four lanes for the other), and instruction k of the group moves row k as ONE 128-byte request; what is
    // loaded that way (B) is transposed back before the arithmetic.  Same sums, same results, bit for bit.
    constexpr bool COAL = UNI && LPR == 2 && NBUF == 2 && !QUAD;
    const bool he = ((sub ^ rot) & 1) != 0;                  // (COAL) this lane keeps accumulator positions 2 m + he: columns of parity `sub`
    auto ocol = [&](int j) { return QUAD ? 4 * qc + ((2 * qh + j + qrot) & 3) : COAL ? (2 * j + (he ? 1 : 0) + rot) & 7 : (rot_out + j) & 7; };
    // (COAL) transposed layout: register k of lane (g = lane >> 3, a = (lane >> 1) & 3, h = lane & 1) belongs to row 4 g + k of the
    // wavefront's 32 and to the column that lane (4 g + k, h) kept at position a
    auto tcol = [&](int k) { return (2 * ((lane >> 1) & 3) + ((lane ^ k) & 1) + ((4 * (lane >> 3) + k) & 7)) & 7; };
    auto transpose4 = [&](cplx (&R)[4]) {                    // 4 x 4 transpose (register index <-> row of the group); its own inverse
        const bool odd = ((lane >> 1) & 1) != 0;
#pragma unroll
        for (int m0 = 0; m0 < 4; m0 += 2) {                  // rows a and a ^ 1: two lanes further on in the quad
            // (component by component: a select between two array ELEMENTS becomes a select of addresses and sends the arrays to scratch)
            const double sx = odd ? R[m0].x : R[m0 + 1].x, sy = odd ? R[m0].y : R[m0 + 1].y;
            const double rx = lane_quad<0x4E>(sx), ry = lane_quad<0x4E>(sy);
            R[m0].x = odd ? rx : R[m0].x;          R[m0].y = odd ? ry : R[m0].y;
            R[m0 + 1].x = odd ? R[m0 + 1].x : rx;  R[m0 + 1].y = odd ? R[m0 + 1].y : ry;
        }
#pragma unroll
        for (int m0 = 0; m0 < 2; ++m0) {                     // rows a and a ^ 2: four lanes further on (row_shl:4 into banks 0, 2; row_shr:4 into banks 1, 3)
            const cplx tmp = R[m0 + 2];
            R[m0 + 2] = cplx{lane_dpp_masked<0x104, 0x5>(R[m0 + 2].x, R[m0].x), lane_dpp_masked<0x104, 0x5>(R[m0 + 2].y, R[m0].y)};
            R[m0] = cplx{lane_dpp_masked<0x114, 0xA>(R[m0].x, tmp.x), lane_dpp_masked<0x114, 0xA>(R[m0].y, tmp.y)};
        }
    };
    const GroupDev G0 = op.g[0];
    const TileGroupDev T0 = td.g0;
    // ---- per-tile state
    int s00, n0, w0, W, r0, nrows;
    int side;                                                // this lane's row in the side-row list (the other groups' part), or -1
    unsigned dslot;                                          // window slot of this lane's own row (the diagonal's column), 0xFFFF: not there
    unsigned ixr[KR];
    double2 avr[KR];
    int gr[NW];                                              // window row list: wave instruction u moves the window rows wv*8 + 64*u .. +7
    // (two steps: the loads are requested in one place and turned into scalars in another, a memory latency later)
    // (read through the constant address space: uniform addresses, so these are scalar loads -- no vector registers held while
    // the answers are under way, and no place in the vector-memory queue)
    typedef const __attribute__((address_space(4))) int *kint_p;
    const kint_p k_sptr = (kint_p)(uintptr_t)T0.sptr, k_win = (kint_p)(uintptr_t)td.win_ptr, k_row = (kint_p)(uintptr_t)td.row_ptr;
    auto request_scalars = [&](int t, int (&raw)[6]) {
        const int sl = QUAD ? t * 8 + (wv >> 1) : t * NWV + wv;     // (QUAD: the storage has 8 slices of 32 rows per tile; a wavefront takes half of one)
        raw[0] = k_sptr[sl]; raw[1] = k_sptr[sl + 1];
        raw[2] = k_win[t]; raw[3] = k_win[t + 1];
        raw[4] = k_row[t]; raw[5] = k_row[t + 1];
    };
    auto take_scalars = [&](const int (&raw)[6], int &s00_, int &n0_, int &w0_, int &W_, int &r0_, int &nrows_) {
        s00_ = __builtin_amdgcn_readfirstlane(raw[0]); n0_ = __builtin_amdgcn_readfirstlane((raw[1] - raw[0]) >> 6);
        w0_ = __builtin_amdgcn_readfirstlane(raw[2]); W_ = __builtin_amdgcn_readfirstlane(raw[3] - raw[2]);
        r0_ = __builtin_amdgcn_readfirstlane(raw[4]); nrows_ = __builtin_amdgcn_readfirstlane(raw[5] - raw[4]);
    };
    auto load_scalars = [&](int t, int &s00_, int &n0_, int &w0_, int &W_, int &r0_, int &nrows_) {
        int raw[6];
        request_scalars(t, raw);
        take_scalars(raw, s00_, n0_, w0_, W_, r0_, nrows_);
    };
    auto load_list = [&](int (&g_)[NW], int w0_, int W_) {
        const int *wl = td.win_cols + w0_;
#pragma unroll
        for (int u = 0; u < NW; ++u) { const int i = wv * 8 + WSTEP * u + (lane >> 3); g_[u] = wl[i < W_ ? i : W_ - 1]; }
    };
    auto load_matrix = [&]() {                               // this lane's share of its row of the bulk group -> registers
#pragma unroll
        for (int u = 0; u < KR; ++u) { ixr[u] = 0; avr[u] = double2{0.0, 0.0}; }
        const unsigned short *__restrict__ si = T0.sidx;
        const double2 *__restrict__ v2 = (const double2 *)T0.svals;
#pragma unroll
        for (int u = 0; u < KR; ++u)
            if (u < n0) { const int e = s00 + (QUAD ? 32 * (wv & 1) + 2 * (lane >> 2) + qh : lane) + 64 * u; ixr[u] = si[e]; avr[u] = v2[e]; }   // (uniform; absent entries stay (0, 0.0))
        const int lr = wv * RPW + lane / LPR;
        side = (td.nside && lr < nrows) ? td.side_of_row[r0 + lr] : -1;
        dslot = (T0.dslot && lr < nrows) ? T0.dslot[r0 + lr] : 0xFFFFu;
    };
    // The LDS-DMA is issued from an asm statement, NOT through __builtin_amdgcn_global_load_lds: hipcc counts the builtin as a
    // pending write to LDS and puts s_waitcnt vmcnt(0) before the next ds_read of ANY address -- the wavefront that had just
    // issued the gather of chunk c+1 sat out its whole HBM round trip before it read chunk c's window (the other buffer).
    // Hidden from that bookkeeping, the gather completes under the compute phase; its completion is waited for explicitly
    // (TILE_DMA_WAIT; vector-memory operations complete in order).  M0 = LDS destination of lane 0, saved and restored in the
    // statement (compiler-reserved register).
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)tile_smem);
    auto glds16 = [&](const cplx *src, unsigned lds_byte) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
    };
#define TILE_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
    // ... or until at most k vector-memory operations are outstanding: the k youngest -- the pieces of the window gathered for the
    // chunk after the next one -- may stay in flight
    auto dma_wait_but = [&](int k) {
        switch (k) {
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    auto pieces_of = [&](int W_) { const int left = W_ - wv * 8; return left <= 0 ? 0 : (left + WSTEP - 1) / WSTEP <= NW ? (left + WSTEP - 1) / WSTEP : 99; };
    // Gather of a window into buffer b, piece by piece (rows past W duplicate the last one into the slack of the 8-row granule:
    // no per-lane predicate).  The pieces of the NEXT window are issued between the entries of the compute phase: a wavefront
    // that issues its ten pieces back to back sits in the issue queue for ~1.2 k cycles.
    auto window_src = [&](int c) {
        int bc = c * 8 + (lane & 7);
        if (bc >= nb) bc = nb - 1;
        return X + bc;
    };
    auto issue_piece = [&](const cplx *Xc, int b, const int (&g_)[NW], int W_, int u) {   // u: compile-time piece number
        const int r = wv * 8 + WSTEP * u;
        if (r < W_) glds16(Xc + (size_t)g_[u] * nb, lds_base + (unsigned)b * (unsigned)wslots * 128u + (unsigned)r * 128u);   // (uniform)
    };
    auto issue_rest = [&](const cplx *Xc, int b, int w0_, int W_) {   // (windows beyond NW sweeps: only with WAE_TILE_WCAP > 640)
        for (int rb = wv * 8 + WSTEP * NW; rb < W_; rb += WSTEP) {
            const int i = rb + (lane >> 3);
            glds16(Xc + (size_t)td.win_cols[w0_ + (i < W_ ? i : W_ - 1)] * nb, lds_base + (unsigned)b * (unsigned)wslots * 128u + (unsigned)rb * 128u);
        }
    };
    auto issue_window = [&](int c, int b, const int (&g_)[NW], int w0_, int W_) {
        const cplx *Xc = window_src(c);
#pragma unroll
        for (int u = 0; u < NW; ++u) issue_piece(Xc, b, g_, W_, u);
        issue_rest(Xc, b, w0_, W_);
    };
    // ---- first tile: scalars, window list, matrix slice; its first window is gathered at the top of the loop (w0 = false)
    load_scalars(tile, s00, n0, w0, W, r0, nrows);
    load_list(gr, w0, W);
    load_matrix();
    int buf = 0;                                             // buffer of the current chunk's window; the windows of the chunks that
    bool wn0 = false, wn1 = false, wn2 = false;              // follow sit in buf + 1, buf + 2 (mod NBUF): wn_k = "has been requested"
    if (spc_all)
        for (int i = tid; i < nch_all * 8 * npl; i += NTHR) {
            const int cc = i / npl, q = i - cc * npl;
            spc0[i] = pc[(size_t)((cc < nb ? cc : nb - 1) / cps) * npl + q];
        }
    // ---- the pipeline over tiles.  The NEXT tile is known from the start of a tile (tile_n: drawn during the tile before); the one
    // after it is drawn during this tile's first chunk (nx = 1: asked, 2: published in LDS, taken over at the switch).  The next
    // tile's scalars are requested at the top of the first chunk and known at its end; its window list replaces this tile's (gr)
    // as soon as this tile has requested its last own window, so that its last chunks gather the next tile's first windows.
    int nx = 0;
    unsigned pend = 0;                                       // (thread 0) the returning atomic of the draw
    bool pub_fresh = false;                                  // published at the end of this very chunk (no barrier since)
    int tile_n = 0, ch_n = 0, ch_end_n = 0;
    bool has_next = false;
    {                                                        // the first "next": drawn here
        if (tid == 0) *vt_slot = draw(-1);
        __syncthreads();
        const int vt = *vt_slot;
        has_next = vt >= 0 && decode(vt, tile_n, ch_n, ch_end_n);
    }
    int s00_n = 0, n0_n = 0, w0_n = 0, W_n = 0, r0_n = 0, nrows_n = 0;
    bool sc_ready = false;                                   // the scalars of tile_n are known
    bool pre_ready = false;                                  // gr holds the window list of tile_n, not this tile's any more
    bool pre_asked = false;                                  // its scalars have been requested (pre_raw)
    int pre_raw[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) pre_raw[u] = 0;
#ifdef WAE_TILE_STAMPS
    if (blockIdx.x == TILE_STAMP_WG && tid == 0) wae_tile_stamps[65] = __builtin_amdgcn_s_memtime();      // prologue done
    int nstamp = 0;
#endif
    while (true) {
        // the chunks that follow: c1 (next), c2 (the one after); in this tile or (in_n) in the next one
        const int c1o = next_active(ch + 1, ch_end);
        const bool same = c1o < ch_end;                      // the next chunk belongs to this tile
        const bool v1 = same || has_next;
        const bool n1 = !same;
        const int c1 = same ? c1o : ch_n;
        int c2 = 0;
        bool v2 = false, n2 = false;
        if (NBUF > 2) {
            if (same) {
                const int c2o = next_active(c1o + 1, ch_end);
                if (c2o < ch_end) { v2 = true; c2 = c2o; }
                else if (has_next) { v2 = true; n2 = true; c2 = ch_n; }
            } else if (has_next) {
                const int c2o = next_active(ch_n + 1, ch_end_n);
                if (c2o < ch_end_n) { v2 = true; n2 = true; c2 = c2o; }
            }
        }
        const int col0 = ch * 8;
        TILE_STAMP(0);
        cplx *spc = spc0 + (spc_all ? (size_t)col0 * npl : 0);
        if (!wn0) {                                          // this chunk's window has not been requested (first chunk of the workgroup,
            if (pre_ready) { load_list(gr, w0, W); pre_ready = false; }      // tiles of very few chunks): gather it now, exposed
            __syncthreads();
            issue_window(ch, buf, gr, w0, W);
            TILE_DMA_WAIT();
            wn0 = true;
        }
        __syncthreads();                                     // everybody's share of this chunk's window has landed (each wavefront
                                                             // waited for its own before its last stores); every wavefront is done
        if (!spc_all) {                                      // with the buffer of the previous chunk and the previous coefficients
            for (int i = tid; i < 8 * npl; i += NTHR) {
                const int cc = i / npl, q = i - cc * npl;
                int bb = col0 + cc;
                if (bb >= nb) bb = nb - 1;
                spc[i] = pc[(size_t)(bb / cps) * npl + q];
            }
            __syncthreads();
        }
        cplx *win = smem + (size_t)buf * wslots * 8;
        TILE_STAMP(1);
        pub_fresh = false;
        if (nx == 0) {                                       // (first chunk of a tile) draw the tile after the next one, and request
            if (has_next) {                                  // the next one's scalars: both answers land under this chunk
                if (tid == 0 && gpx < share_size(xcd)) pend = atomicAdd(td.counters + xcd, 1u);
                request_scalars(tile_n, pre_raw);
                pre_asked = true;
            }
            nx = 1;
        }
        // this lane's rows and columns in the epilogue
        const int lrow = wv * RPW + lane / LPR;              // this lane's row inside the tile
        const bool live = lrow < nrows;
        const int64_t row = r0 + (live ? lrow : 0);
        const bool need_b = mode == MODE_RES || mode == MODE_ADD || mode == MODE_RES_DS || mode == MODE_JAC;
        const bool need_d = !(mode == MODE_AX || mode == MODE_RES || mode == MODE_ADD);
        const double dsg = op.conj_diag ? -1.0 : 1.0;
        cplx bv[NOUT], xv[NOUT];
        cplx dgu = {1.0, 0.0};
        auto trow = [&](int k) { return wv * RPW + 4 * (lane >> 3) + k; };       // (COAL) row of register k in the transposed layout
        auto b_loads = [&]() {                               // (COAL) B row by row, 128 bytes per 8-lane group and instruction
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int lr = trow(k) < nrows ? trow(k) : (nrows > 0 ? nrows - 1 : 0);
                const int c = tcol(k);
                const int b = col0 + c < nb ? col0 + c : nb - 1;
                bv[k] = need_b ? B[(size_t)(r0 + lr) * nb + b] : cplx{0.0, 0.0};
            }
        };
        auto epilogue_loads = [&]() {                        // right-hand sides of the fused modes: requested together
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                const int c = ocol(j);
                const int b = col0 + c < nb ? col0 + c : nb - 1;
                const size_t e = (size_t)row * nb + b;
                if constexpr (!COAL) bv[j] = need_b ? B[e] : cplx{0.0, 0.0};
                xv[j] = (mode == MODE_JAC && dslot == 0xFFFFu) ? X[e] : cplx{0.0, 0.0};       // (normally taken from the window, below)
            }
            if (UNI && need_d) {                             // one diagonal per row and chunk
                dgu = cplx{0.0, 0.0};
                for (int q = 0; q < npl; ++q) { cplx dq = op.diag[(size_t)row * npl + q]; dq.y *= dsg; cfma(dgu, spc[q], dq); }
            }
        };
        // With three buffers the window gathered under this chunk belongs to the chunk after the next one and must stay in flight
        // past this chunk's end: whatever the epilogue loads from memory is requested HERE, ahead of the window's pieces (vector-
        // memory operations complete in order, and the compiler's wait for a load retires everything older with it).  The
        // right-hand side goes straight into the accumulators (positions this lane will keep, sign such that they end up holding
        // A x - b, or A x + b for MODE_ADD): no registers of its own.
        // SPLIT (the columns of a chunk belong to several systems -- a rank's share of the probe columns in a multi-GPU pass, the
        // Newton-type solvers): the two planes are accumulated separately, acc = M-plane x X and acc2 = K-plane x X, real times
        // complex, and the systems' coefficients are applied once per chunk in the epilogue.  (Forming c0 m + c1 k per entry and
        // COLUMN cost 16 more LDS reads and 28 more FMAs per entry: 1 150 us against 760 at 1M unknowns, 64 columns.)
        constexpr bool SPLIT = !UNI;
        constexpr bool FOLD = NBUF > 2 && !SPLIT;            // the right-hand side starts out in the accumulators
        cplx acc[8], acc2[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) { acc[s] = cplx{0.0, 0.0}; acc2[s] = cplx{0.0, 0.0}; }
        // (COAL) the right-hand side is requested HERE, ahead of the window's pieces: it lands under the compute phase instead of
        // holding the epilogue -- 16 more registers alive through the entries (256 in all, no scratch); residual 885 -> 850 us at 1M
        // unknowns and 64 columns
        if constexpr (COAL) b_loads();
        if (NBUF > 2) {
            epilogue_loads();
            const double sb = mode == MODE_ADD ? 1.0 : -1.0;
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                if (!FOLD) continue;
                if (LPR == 2) {
                    acc[j] = cplx{sub == 0 ? sb * bv[j].x : 0.0, sub == 0 ? sb * bv[j].y : 0.0};
                    acc[j + 4] = cplx{sub == 1 ? sb * bv[j].x : 0.0, sub == 1 ? sb * bv[j].y : 0.0};
                } else {
                    acc[j] = cplx{sb * bv[j].x, sb * bv[j].y};
                }
            }
        }
        // windows to request under this chunk: the one NBUF - 1 chunks ahead; with three buffers also the next one, should it be
        // missing (start of the pipeline).  A window of this tile needs this tile's list in gr, one of the next tile the other list.
        if (NBUF > 2 && !wn1 && v1 && (n1 ? pre_ready : !pre_ready)) {
            issue_window(c1, (buf + 1) % NBUF, gr, n1 ? w0_n : w0, n1 ? W_n : W);
            wn1 = true;
        }
        const bool vt_ = NBUF > 2 ? v2 : v1, nt_ = NBUF > 2 ? n2 : n1;
        const bool already = NBUF > 2 ? wn2 : wn1;
        const bool more = vt_ && !already && (nt_ ? pre_ready : !pre_ready);     // a window is gathered under this chunk
        const int ct = NBUF > 2 ? c2 : c1;
        const cplx *Xn = window_src(more ? ct : ch);
        const int Wd = nt_ ? W_n : W;                        // rows of that window
        const int bt = (buf + NBUF - 1) % NBUF;              // its buffer
        TILE_STAMP(2);
        // SPLIT on two lanes per row: two passes over the entries, four columns each (half the accumulators and operands at a time:
        // with all eight columns the second set of accumulators did not fit the registers, 208 B of spill in the loop)
        constexpr bool HALVES = SPLIT && LPR == 2;
        cplx res[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; ++j) res[j] = cplx{0.0, 0.0};
        if (HALVES && n0 > 0) {
            const unsigned short *__restrict__ si = T0.sidx;
            const double2 *__restrict__ v2 = (const double2 *)T0.svals;
            auto half_pass = [&](auto HC) {
                constexpr int H = decltype(HC)::value;
                cplx p[4], q[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) { p[s] = cplx{0.0, 0.0}; q[s] = cplx{0.0, 0.0}; }
                auto fetch4 = [&](cplx (&x)[4], unsigned ix) {
                    const cplx *wr = win + ix * 8;
#pragma unroll
                    for (int s = 0; s < 4; ++s) x[s] = wr[(4 * H + s + rot) & 7];
                };
                auto apply4 = [&](const cplx (&x)[4], double2 a) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        p[s].x = fma(a.x, x[s].x, p[s].x); p[s].y = fma(a.x, x[s].y, p[s].y);
                        q[s].x = fma(a.y, x[s].x, q[s].x); q[s].y = fma(a.y, x[s].y, q[s].y);
                    }
                };
                unsigned ixs = 0;
                double2 avs = {0.0, 0.0};
                if (n0 > KR) { const int e = s00 + lane + 64 * KR; ixs = si[e]; avs = v2[e]; }
                cplx xa[4], xb[4];
                fetch4(xa, ixr[0]);
#pragma unroll
                for (int u = 0; u < KR; ++u) {
                    if (u + 1 < KR) { if (u & 1) fetch4(xa, ixr[u + 1]); else fetch4(xb, ixr[u + 1]); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (u & 1) apply4(xb, avr[u]); else apply4(xa, avr[u]);
                    asm volatile("" : "+v"(p[0].x), "+v"(p[0].y), "+v"(p[1].x), "+v"(p[1].y), "+v"(p[2].x), "+v"(p[2].y), "+v"(p[3].x), "+v"(p[3].y),
                                      "+v"(q[0].x), "+v"(q[0].y), "+v"(q[1].x), "+v"(q[1].y), "+v"(q[2].x), "+v"(q[2].y), "+v"(q[3].x), "+v"(q[3].y)
                                 : : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    if (H == 0 && more && 2 * u < NW) {
                        issue_piece(Xn, bt, gr, Wd, 2 * u);
                        if (2 * u + 1 < NW) issue_piece(Xn, bt, gr, Wd, 2 * u + 1);
                    }
                }
                if (H == 0 && more) issue_rest(Xn, bt, nt_ ? w0_n : w0, Wd);
#pragma unroll 1
                for (int k = KR; k < n0; ++k) {
                    const unsigned ixc = ixs;
                    const double2 avc = avs;
                    if (k + 1 < n0) { const int e = s00 + lane + 64 * (k + 1); ixs = si[e]; avs = v2[e]; }
                    cplx x[4];
                    fetch4(x, ixc);
                    apply4(x, avc);
                }
                // the two lanes of a row meet; lane `sub` == H keeps these four columns: (rot_out + j) mod 8 for it
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const cplx P = {p[j].x + lane_quad<0xB1>(p[j].x), p[j].y + lane_quad<0xB1>(p[j].y)};
                    const cplx Q = {q[j].x + lane_quad<0xB1>(q[j].x), q[j].y + lane_quad<0xB1>(q[j].y)};
                    const int cs = (4 * H + j + rot) & 7;
                    const cplx d0 = spc[cs * npl + G0.plane0], d1 = spc[cs * npl + G0.plane0 + 1];
                    cplx t = {0.0, 0.0};
                    cfma(t, d0, P);
                    cfma(t, d1, Q);
                    if (sub == H) res[j] = t;
                }
            };
            half_pass(std::integral_constant<int, 0>{});
            half_pass(std::integral_constant<int, 1>{});
        } else if (HALVES) {
            if (more) issue_window(ct, bt, gr, nt_ ? w0_n : w0, Wd);
        } else
        if (n0 > 0) {                                        // the bulk group: mass + stiffness on one pattern, 16 B + 2 B per nonzero
            const cplx c0 = td.unit ? cplx{1.0, 0.0} : spc[G0.plane0], c1p = td.unit ? cplx{0.0, 0.0} : spc[G0.plane0 + 1];
            auto fetch = [&](cplx (&x)[8], unsigned ix) {    // the 8 operands of one entry: 8 ds_read_b128, rotated column order
                const cplx *wr = win + ix * 8;               // (QUAD: the four of this lane's column half)
#pragma unroll
                for (int s = 0; s < (QUAD ? 4 : 8); ++s) x[s] = wr[QUAD ? 4 * qc + ((s + qrot) & 3) : (s + rot) & 7];
            };
            auto apply = [&](const cplx (&x)[8], double2 a) {
                if (SPLIT) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        acc[s].x = fma(a.x, x[s].x, acc[s].x); acc[s].y = fma(a.x, x[s].y, acc[s].y);
                        acc2[s].x = fma(a.y, x[s].x, acc2[s].x); acc2[s].y = fma(a.y, x[s].y, acc2[s].y);
                    }
                } else {
                    const cplx m = {fma(c0.x, a.x, c1p.x * a.y), fma(c0.y, a.x, c1p.y * a.y)};
#pragma unroll
                    for (int s = 0; s < (QUAD ? 4 : 8); ++s) cfma(acc[s], m, x[s]);
                }
            };
            const unsigned short *__restrict__ si = T0.sidx;
            const double2 *__restrict__ v2 = (const double2 *)T0.svals;
            // rows longer than LPR * KR entries: the rest is streamed from L2, one entry ahead; the first of them is requested here,
            // ahead of the window pieces (vector-memory operations complete in order)
            unsigned ixs = 0;
            double2 avs = {0.0, 0.0};
            const int mlane = QUAD ? 32 * (wv & 1) + 2 * (lane >> 2) + qh : lane;      // this lane's place in its slice of the storage
            if (n0 > KR) { const int e = s00 + mlane + 64 * KR; ixs = si[e]; avs = v2[e]; }
            // register-resident part, software-pipelined by hand: the operands of entry u+1 are requested before the FMAs of
            // entry u, two operand sets alternate; absent entries are (slot 0, 0.0).  The scheduling fences pin that order --
            // left alone hipcc hoists 32 reads, runs out of registers and then issues the rest two at a time behind
            // s_waitcnt lgkmcnt(0).
            cplx xa[8], xb[8];
            fetch(xa, ixr[0]);
#pragma unroll
            for (int u = 0; u < KR; ++u) {
                if constexpr (!QUAD) {
                    if (u + 1 < KR) { if (u & 1) fetch(xa, ixr[u + 1]); else fetch(xb, ixr[u + 1]); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (u & 1) apply(xb, avr[u]); else apply(xa, avr[u]);
                } else {                                     // (one operand set: four wavefronts per SIMD cover the LDS latency)
                    if (u > 0) fetch(xa, ixr[u]);
                    apply(xa, avr[u]);
                }
                // (the FMAs are pure arithmetic: only an operand dependence keeps them from sinking below the later reads)
                if constexpr (QUAD)
                    asm volatile("" : "+v"(acc[0].x), "+v"(acc[0].y), "+v"(acc[1].x), "+v"(acc[1].y), "+v"(acc[2].x), "+v"(acc[2].y),
                                      "+v"(acc[3].x), "+v"(acc[3].y) : : "memory");
                else
                asm volatile("" : "+v"(acc[0].x), "+v"(acc[0].y), "+v"(acc[1].x), "+v"(acc[1].y), "+v"(acc[2].x), "+v"(acc[2].y),
                                  "+v"(acc[3].x), "+v"(acc[3].y), "+v"(acc[4].x), "+v"(acc[4].y), "+v"(acc[5].x), "+v"(acc[5].y),
                                  "+v"(acc[6].x), "+v"(acc[6].y), "+v"(acc[7].x), "+v"(acc[7].y) : : "memory");
                if (SPLIT)
                    asm volatile("" : "+v"(acc2[0].x), "+v"(acc2[0].y), "+v"(acc2[1].x), "+v"(acc2[1].y), "+v"(acc2[2].x), "+v"(acc2[2].y),
                                      "+v"(acc2[3].x), "+v"(acc2[3].y), "+v"(acc2[4].x), "+v"(acc2[4].y), "+v"(acc2[5].x), "+v"(acc2[5].y),
                                      "+v"(acc2[6].x), "+v"(acc2[6].y), "+v"(acc2[7].x), "+v"(acc2[7].y) : : "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (more && 2 * u < NW) {                    // the window, two pieces per entry: all under way by the middle of the phase
                    issue_piece(Xn, bt, gr, Wd, 2 * u);
                    if (2 * u + 1 < NW) issue_piece(Xn, bt, gr, Wd, 2 * u + 1);
                }
            }
            if (more) issue_rest(Xn, bt, nt_ ? w0_n : w0, Wd);
#pragma unroll 1
            for (int k = KR; k < n0; ++k) {
                const unsigned ixc = ixs;
                const double2 avc = avs;
                if (k + 1 < n0) { const int e = s00 + mlane + 64 * (k + 1); ixs = si[e]; avs = v2[e]; }
                cplx x[8];
                fetch(x, ixc);
                apply(x, avc);
            }
        } else if (more) {
            issue_window(ct, bt, gr, nt_ ? w0_n : w0, Wd);
        }
        if (more) { if (NBUF > 2) wn2 = true; else wn1 = true; }
        TILE_STAMP(3);
        // the partial sums of a row meet; afterwards each lane keeps NOUT results: those of the columns (rot_out + j) mod 8
        auto meet = [&](const cplx (&a)[8], cplx (&r)[NOUT]) {
            if constexpr (QUAD) {                            // the two entry halves of a (row, column half) meet; lane h keeps positions 2 h, 2 h + 1
#pragma unroll
                for (int j = 0; j < NOUT; ++j) {
                    const cplx lo = {a[j].x + lane_quad<0xB1>(a[j].x), a[j].y + lane_quad<0xB1>(a[j].y)};
                    const cplx hi = {a[j + 2].x + lane_quad<0xB1>(a[j + 2].x), a[j + 2].y + lane_quad<0xB1>(a[j + 2].y)};
                    r[j] = qh ? hi : lo;
                }
            } else
            if (LPR == 2) {
#pragma unroll
                for (int j = 0; j < NOUT; ++j) {
                    const cplx lo = {a[j].x + lane_quad<0xB1>(a[j].x), a[j].y + lane_quad<0xB1>(a[j].y)};
                    const cplx hi = {a[j + 4].x + lane_quad<0xB1>(a[j + 4].x), a[j + 4].y + lane_quad<0xB1>(a[j + 4].y)};
                    r[j] = sub ? hi : lo;
                }
            } else {                                         // lane q takes positions j + 2 d from lane q - d, d = 0..3
#pragma unroll
                for (int j = 0; j < NOUT; ++j)
                    r[j] = cplx{a[j].x + lane_quad<0x93>(a[j + 2].x) + lane_quad<0x4E>(a[j + 4].x) + lane_quad<0x39>(a[j + 6].x),
                                a[j].y + lane_quad<0x93>(a[j + 2].y) + lane_quad<0x4E>(a[j + 4].y) + lane_quad<0x39>(a[j + 6].y)};
            }
        };
        if constexpr (COAL) {                                // the two lanes of a row meet; lane `sub` keeps the columns of its parity
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const double sx = he ? acc[2 * m].x : acc[2 * m + 1].x, sy = he ? acc[2 * m].y : acc[2 * m + 1].y;
                const double mx = he ? acc[2 * m + 1].x : acc[2 * m].x, my = he ? acc[2 * m + 1].y : acc[2 * m].y;
                res[m] = cplx{mx + lane_quad<0xB1>(sx), my + lane_quad<0xB1>(sy)};
            }
        } else if (!HALVES) meet(acc, res);
        if (SPLIT && !HALVES) {                              // res = c0 (M x) + c1 (K x), the coefficients of each column's system
            cplx res2[NOUT];
            meet(acc2, res2);
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                const int cs = (rot_out + j) & 7;
                const cplx d0 = td.unit ? cplx{1.0, 0.0} : spc[cs * npl + G0.plane0], d1 = td.unit ? cplx{0.0, 0.0} : spc[cs * npl + G0.plane0 + 1];
                cplx t = {0.0, 0.0};
                cfma(t, d0, res[j]);
                cfma(t, d1, res2[j]);
                res[j] = t;
            }
        }
        TILE_STAMP(4);
        // Epilogue.  (Two buffers: loads first, then the wait for this wavefront's pieces of the next window, which sits before the
        // stores: those drain under the next chunk.)
        if (NBUF == 2) epilogue_loads();
        if constexpr (COAL) { if (need_b) transpose4(bv); }  // (back to this lane's own row: position m <-> column ocol(m))
        if (mode == MODE_JAC && dslot != 0xFFFFu) {          // the row's own X: it is in the window (1 GB per sweep not read again)
#pragma unroll
            for (int j = 0; j < NOUT; ++j) xv[j] = win[dslot * 8 + ocol(j)];
        }
        if (side >= 0) {                                     // the other groups' part of this row (spmv_side_kernel)
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                const int c = ocol(j);
                const int b = col0 + c < nb ? col0 + c : nb - 1;
                const cplx sv = td.side_acc[(size_t)side * nb + b];
                res[j].x += sv.x; res[j].y += sv.y;
            }
        }
        cplx out[NOUT], b2[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
            const int c = ocol(j);
            const cplx av = res[j];
            b2[j] = cplx{0.0, 0.0};
            // (three buffers: av is A x - b, or A x + b, already: bz stands in for the right-hand side)
            const cplx bz = FOLD ? cplx{0.0, 0.0} : bv[j];
            if (mode == MODE_AX) {
                out[j] = av;
            } else if (mode == MODE_RES) {
                out[j] = cplx{bz.x - av.x, bz.y - av.y};
            } else if (mode == MODE_ADD) {
                out[j] = cplx{bz.x + av.x, bz.y + av.y};
            } else {
                cplx dg = dgu;
                if (!UNI) {
                    const cplx *mypc = spc + c * npl;
                    dg = cplx{0.0, 0.0};
                    for (int q = 0; q < npl; ++q) { cplx dq = op.diag[(size_t)row * npl + q]; dq.y *= dsg; cfma(dg, mypc[q], dq); }
                }
                if (mode == MODE_AX_J0) {
                    out[j] = av;
                    const cplx r = cdiv(av, dg);
                    b2[j] = cplx{jac_w * r.x, jac_w * r.y};
                } else if (mode == MODE_AX_DS) {
                    out[j] = cdiv(av, dg);
                } else if (mode == MODE_RES_DS) {
                    out[j] = cdiv(cplx{bz.x - av.x, bz.y - av.y}, dg);
                } else {
                    const cplx r = cdiv(cplx{bz.x - av.x, bz.y - av.y}, dg);
                    out[j] = cplx{xv[j].x + jac_w * r.x, xv[j].y + jac_w * r.y};
                }
            }
        }
        TILE_STAMP(5);
        // this wavefront's pieces of the NEXT chunk's window must have landed (the barrier at the top of the next chunk tells the
        // others); with three buffers the pieces just issued -- the youngest operations -- stay in flight
        if (NBUF > 2 && more) dma_wait_but(pieces_of(Wd)); else TILE_DMA_WAIT();
        if (nx == 1) {                                       // the draw has come back: publish it (read at the tile switch)
            if (tid == 0 && has_next) *vt_slot = draw(gpx < share_size(xcd) ? gpx + (int)pend : -1);
            nx = 2;
            pub_fresh = true;
        }
        if (pre_asked) {
            take_scalars(pre_raw, s00_n, n0_n, w0_n, W_n, r0_n, nrows_n);
            pre_asked = false;
            sc_ready = true;
        }
        // this tile has requested its last own window once fewer than NBUF - 1 of its chunks follow the next one: from then on gr
        // holds the next tile's list (requested here, no wait: used under the next chunk at the earliest)
        if (sc_ready && !pre_ready && has_next && same) {
            int left = 0;                                    // own chunks after c1
            for (int c = next_active(c1o + 1, ch_end); c < ch_end && left < NBUF; c = next_active(c + 1, ch_end)) ++left;
            if (left < NBUF - 1) { load_list(gr, w0_n, W_n); pre_ready = true; }
        }
        if constexpr (COAL) {                                // row by row: instruction k of an 8-lane group writes the 128 bytes of row k
            transpose4(out);
            if (mode == MODE_AX_J0) transpose4(b2);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b = col0 + tcol(k);
                if (trow(k) >= nrows || b >= nb) continue;
                const size_t e = (size_t)(r0 + trow(k)) * nb + b;
                // (streaming stores, `nt`: a gigabyte per product that nothing in this launch reads again should not push the windows
                // out of the 4-MB L2s; product + first sweep 954 -> 939 us)
                typedef double dv2_t __attribute__((ext_vector_type(2)));
                __builtin_nontemporal_store(dv2_t{out[k].x, out[k].y}, (dv2_t *)(Y + e));
                if (mode == MODE_AX_J0) __builtin_nontemporal_store(dv2_t{b2[k].x, b2[k].y}, (dv2_t *)(const_cast<cplx *>(B) + e));
            }
        } else
        if (live) {
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                const int b = col0 + ocol(j);
                if (b >= nb) continue;
                const size_t e = (size_t)row * nb + b;
                Y[e] = out[j];
                if (mode == MODE_AX_J0) const_cast<cplx *>(B)[e] = b2[j];
            }
        }
        TILE_STAMP(6);
#ifdef WAE_TILE_STAMPS
        ++nstamp;
#endif
        buf = (buf + 1) % NBUF;
        wn0 = wn1; wn1 = wn2; wn2 = false;
        if (same) {
            ch = c1o;
        } else {                                             // ---- tile switch
            if (!has_next) break;
            if (!pre_ready) {                                // (tiles of very few chunks: the list could not be prefetched)
                if (!sc_ready) load_scalars(tile_n, s00_n, n0_n, w0_n, W_n, r0_n, nrows_n);
                load_list(gr, w0_n, W_n);
            }
            tile = tile_n; ch = ch_n; ch_end = ch_end_n;
            s00 = s00_n; n0 = n0_n; w0 = w0_n; W = W_n; r0 = r0_n; nrows = nrows_n;
            load_matrix();                                   // (lands under the barrier and the first reads of the next chunk)
            pre_ready = false;
            sc_ready = false;
            if (pub_fresh) __syncthreads();                  // (single-chunk tiles: published a moment ago)
            {
                const int vt = *vt_slot;
                has_next = vt >= 0 && decode(vt, tile_n, ch_n, ch_end_n);
            }
            nx = 0;
        }
    }
    leave();
#ifdef WAE_TILE_STAMPS
    if (blockIdx.x == TILE_STAMP_WG && tid == 0) { TILE_DMA_WAIT(); wae_tile_stamps[66] = __builtin_amdgcn_s_memtime(); }   // last stores done
    if (tid == 0 && blockIdx.x < 1024) { wae_tile_wglog[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime(); wae_tile_wglog[blockIdx.x * 4 + 2] = nstamp; }
#endif
}

#ifdef WAE_TILE_STAMPS
extern "C" int wae_debug_tile_wglog(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wae_tile_wglog), sizeof(unsigned long long) * 1024 * 4);
}
extern "C" int wae_debug_tile_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wae_tile_stamps), sizeof(unsigned long long) * (8 * 64 + 8));
}
#endif
static void launch_spmv_tile(const OpDev &op, const TileDev &td, const cplx *pc, int cps, const cplx *X, cplx *Y, const cplx *B, double jac_w,
                             int nb, int mode, hipStream_t st, const unsigned char *cmask) {
    // (the opt-in to more than 64 KB of dynamic LDS is a property of the function ON A DEVICE: one process may drive several --
    // wae_beyn_moments_mgpu runs a host thread per device -- so it is kept per device, under a lock)
    static std::mutex attr_mutex;
    static bool attr_done[64] = {false};
    int dev_now = 0;
    HIP_CHECK(hipGetDevice(&dev_now));
    const int nbuf = td.nbuf == 3 && td.lpr == 2 ? 3 : 2;
    const size_t wslots = (size_t)((td.wmax + 7) & ~7);
    const int nch8 = ((nb + 7) / 8) * 8;
    size_t shm = nbuf * wslots * 8 * sizeof(cplx) + (size_t)nch8 * op.nplanes_total * sizeof(cplx) + 16;   // coefficients of all chunks staged once ...
    const int spc_all = shm <= 160 * 1024;                   // (+ 16: the word through which a workgroup learns its next tile)
    if (!spc_all) shm = nbuf * wslots * 8 * sizeof(cplx) + (size_t)8 * op.nplanes_total * sizeof(cplx) + 16;   // ... or chunk by chunk
    if (shm > 160 * 1024) throw WaeError(WAE_ERR_INVALID, "tile windows do not fit LDS (WAE_TILE_WCAP too large)");
    std::unique_lock<std::mutex> attr_lock(attr_mutex);
    bool &attr_set = attr_done[dev_now & 63];
    if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void *)spmv_tile_kernel<true, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void *)spmv_tile_kernel<false, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void *)spmv_tile_kernel<true, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void *)spmv_tile_kernel<false, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void *)spmv_tile_kernel<true, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void *)spmv_tile_kernel<false, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void *)spmv_tile_kernel<true, 4, 2, 16, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    static int ncu_dev[64] = {0};                            // CUs per device (devices of a node may differ; under the lock)
    int &ncu_slot = ncu_dev[dev_now & 63];
    if (!ncu_slot) {
        HIP_CHECK(hipDeviceGetAttribute(&ncu_slot, hipDeviceAttributeMultiprocessorCount, dev_now));
        if (getenv("WAE_TILE_GRID")) ncu_slot = atoi(getenv("WAE_TILE_GRID"));
        ncu_slot = ncu_slot < 8 ? 8 : ncu_slot & ~7;
    }
    const int ncu = ncu_slot;
    attr_lock.unlock();
    // Persistent workgroups, one per CU (157 KB of LDS each: the hardware cannot place two on a CU), walking the tiles of
    // their XCD.  With fewer tiles than CUs the chunks of a tile are shared out between csplit workgroups.  (Sharing them out
    // as soon as a CU had fewer than 8 tiles was measured and is slower -- 200k DoF, 780 tiles: 229 -> 259 us: the matrix
    // slice is re-loaded and the window pipeline restarts per part.)
    const int nchunks = (nb + 7) / 8;
    // parts in which the last tiles of every XCD's share are handed out (kernel: "Work list"): with fewer tiles than CUs all of
    // them, so that every CU has work; otherwise four parts bring the end of the launch within a quarter of a tile on every CU
    // (parts of at least two chunks: a one-chunk part cannot prefetch its successor.  Measured, r = 64: 200k unknowns 210 -> 192 us,
    // 1M unknowns 764 -> 755 us)
    int csplit = td.ntiles < ncu ? (int)((ncu + td.ntiles - 1) / td.ntiles)
                                 : std::min(getenv("WAE_TILE_TAIL") ? atoi(getenv("WAE_TILE_TAIL")) : 4, std::max(1, nchunks / 2));
    if (csplit > nchunks) csplit = nchunks;
    if (csplit < 1) csplit = 1;
    const unsigned tpx = (unsigned)((td.ntiles + 7) / 8);
    const unsigned units = std::min<unsigned>(tpx, (unsigned)(ncu / 8)) * (unsigned)csplit + (tpx > (unsigned)(ncu / 8) ? tpx - (unsigned)(ncu / 8) : 0u);
    const dim3 grid(8u * std::min(units, (unsigned)(ncu / 8)));
    if (td.nside) {
        if (nb > 256) throw WaeError(WAE_ERR_INVALID, "launch_spmv: batch wider than 256 columns");
        hipLaunchKernelGGL(spmv_side_kernel, dim3((unsigned)((td.nside + 31) / 32), (unsigned)nchunks), dim3(256), 0, st, td, op.nplanes_total,
                           op.conj_diag, pc, cps, X, nb, cmask);
        HIP_CHECK(hipGetLastError());
        if (td.nlong_side) {
            hipLaunchKernelGGL(spmv_side_long_kernel, dim3((unsigned)td.nlong_side * WAE_LONG_SPLIT, (unsigned)nchunks), dim3(256), 0, st, td,
                               op.nplanes_total, op.conj_diag, pc, cps, X, nb, cmask);
            HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL(long_reduce_kernel, dim3((unsigned)((td.nlong_side * nb + 255) / 256)), dim3(256), 0, st, td.ls_part, td.nlong_side, nb,
                               td.ls_side, td.side_acc, cmask);
            HIP_CHECK(hipGetLastError());
        }
    }
#define WAE_TILE_LAUNCH(U, L, N) hipLaunchKernelGGL((spmv_tile_kernel<U, L, N>), grid, dim3(512), shm, st, op, td, pc, cps, X, Y, B, jac_w, nb, mode, cmask, spc_all, csplit)
    // QUAD form (16 wavefronts, four lanes per row): the same tile storage as the 2-lane form; one system per chunk only
    static const int waves16 = getenv("WAE_TILE_WAVES") ? atoi(getenv("WAE_TILE_WAVES")) == 16 : 0;
    if (waves16 && td.lpr == 2 && nbuf == 2 && cps % 8 == 0 && !td.unit)
        hipLaunchKernelGGL((spmv_tile_kernel<true, 4, 2, 16, 8>), grid, dim3(1024), shm, st, op, td, pc, cps, X, Y, B, jac_w, nb, mode, cmask, spc_all, csplit);
    else
    if (td.lpr == 4) { if (cps % 8 == 0) WAE_TILE_LAUNCH(true, 4, 2); else WAE_TILE_LAUNCH(false, 4, 2); }
    else if (nbuf == 3) { if (cps % 8 == 0) WAE_TILE_LAUNCH(true, 2, 3); else WAE_TILE_LAUNCH(false, 2, 3); }
    else { if (cps % 8 == 0) WAE_TILE_LAUNCH(true, 2, 2); else WAE_TILE_LAUNCH(false, 2, 2); }
#undef WAE_TILE_LAUNCH
    HIP_CHECK(hipGetLastError());
}

typedef void (*spmv_fn)(OpDev, const cplx *, int, const cplx *, cplx *, const cplx *, double, int, int, const unsigned char *);
template <int C, int S> static spmv_fn spmv_ptr() { return spmv_kernel<C, S>; }

static spmv_fn pick_spmv(int C, int S) {
#define PICK(c, s_) if (C == c && S == s_) return spmv_ptr<c, s_>();
    PICK(1, 1) PICK(1, 2) PICK(1, 4) PICK(1, 8) PICK(1, 16) PICK(1, 32)
    PICK(2, 1) PICK(2, 2) PICK(2, 4) PICK(2, 8) PICK(2, 16)
    PICK(4, 1) PICK(4, 2) PICK(4, 4) PICK(4, 8) PICK(4, 16)
    PICK(8, 1) PICK(8, 2) PICK(8, 4) PICK(8, 8)
    PICK(16, 1) PICK(16, 2) PICK(16, 4)
#undef PICK
    return nullptr;
}

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

void launch_spmv(const OpDev &op, const cplx *pc, int cps, const cplx *X, cplx *Y, const cplx *B, double jac_w,
                 int nb, int mode, hipStream_t st, const unsigned char *cmask) {
    const int envC = env_int("WAE_SPMV_C", 0), envS = env_int("WAE_SPMV_S", 0);   // tuning overrides
    int C = nb >= 8 ? 8 : (nb >= 4 ? 4 : (nb >= 2 ? 2 : 1));
    int S = 64 / C >= 8 ? 8 : 64 / C;
    if (C == 8) S = 1;
    // small (coarse-level / transfer) operators are latency-bound with one lane per row: split each row over S lanes
    if (C == 8 && op.n < env_int("WAE_SPMV_SMALL8", 4096)) S = 8;
    else if (C == 8 && op.n < env_int("WAE_SPMV_SMALL4", 65536)) S = 4;
    // narrow batches (the Newton-type solvers; no tile path below 8 columns) on a large operator: teams of 8 lanes -- with 15-48 entries
    // per row more lanes per row leave most of a team idle in the reduction (4 columns at 1M unknowns, solve of 26 iterations: 8 / 4 / 2 /
    // 1 lanes per row 66 / 51 / 47 / 53 ms; 2 columns 35 / 29 / 31 / 39 ms; 1 column 17.5 / 18.5 / 22.6 / 32 ms)
    else if (C < 8 && op.n >= env_int("WAE_SPMV_SMALL4", 65536)) S = 8 / C;
    if (envC > 0 && envC <= nb) C = envC;
    if (envS > 0) S = envS;
    if (op.n <= 0) return;
    if (op.tiles && nb >= 8 && envC == 0 && envS == 0 && env_int("WAE_SPMV_TILE", 1)) {
        launch_spmv_tile(op, *op.tiles, pc, cps, X, Y, B, jac_w, nb, mode, st, cmask);
        return;
    }
    if (op.nlong) {
        if (nb > 256) throw WaeError(WAE_ERR_INVALID, "launch_spmv: batch wider than 256 columns");
        launch_long_rows(op, pc, cps, X, nb, nb >= 8 ? cmask : (const unsigned char *)nullptr, st);
    }
    if (mode == MODE_AX_J0 && !(C == 8 && S == 1 && env_int("WAE_SPMV_LDS", 1))) {   // only the wide fine-level kernel fuses the sweep
        launch_spmv(op, pc, cps, X, Y, nullptr, 0.0, nb, MODE_AX, st, cmask);
        launch_jacobi0(op, pc, cps, Y, const_cast<cplx *>(B), jac_w, nb, st, cmask);
        return;
    }
    if (C == 8 && S == 1 && env_int("WAE_SPMV_LDS", 1)) {
        const int nch_env = env_int("WAE_SPMV_NCH", 0);
        int nchunks = (nb + 7) / 8;
        int NCH = nch_env > 0 ? nch_env : (nchunks >= 4 ? 4 : (nchunks >= 2 ? 2 : 1));   // 8 halves the occupancy: slower (measured)
        const unsigned nrb = (unsigned)((op.n + 31) / 32);
        dim3 grid((nrb + 7u) / 8u * 8u, (unsigned)((nb + 8 * NCH - 1) / (8 * NCH)));
        size_t shm = (size_t)8 * NCH * op.nplanes_total * sizeof(cplx);
        if (NCH == 8) hipLaunchKernelGGL(spmv_lds_kernel<8>, grid, dim3(256), shm, st, op, pc, cps, X, Y, B, jac_w, nb, mode, cmask);
        else if (NCH == 4) hipLaunchKernelGGL(spmv_lds_kernel<4>, grid, dim3(256), shm, st, op, pc, cps, X, Y, B, jac_w, nb, mode, cmask);
        else if (NCH == 2) hipLaunchKernelGGL(spmv_lds_kernel<2>, grid, dim3(256), shm, st, op, pc, cps, X, Y, B, jac_w, nb, mode, cmask);
        else hipLaunchKernelGGL(spmv_lds_kernel<1>, grid, dim3(256), shm, st, op, pc, cps, X, Y, B, jac_w, nb, mode, cmask);
        HIP_CHECK(hipGetLastError());
        return;
    }
    spmv_fn fn = pick_spmv(C, S);
    if (!fn) throw WaeError(WAE_ERR_INVALID, "launch_spmv: unsupported (C,S)");
    const int tpb = 256 / (C * S);
    const unsigned nrb = (unsigned)((op.n + tpb - 1) / tpb);
    dim3 grid((nrb + 7u) / 8u * 8u, (unsigned)((nb + C - 1) / C));
    size_t shm = (size_t)C * op.nplanes_total * sizeof(cplx);
    hipLaunchKernelGGL(fn, grid, dim3(256), shm, st, op, pc, cps, X, Y, B, jac_w, nb, mode, C == 8 ? cmask : (const unsigned char *)nullptr);
    HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void jacobi0_kernel(OpDev op, const cplx *__restrict__ pc, int cps,
                                                      const cplx *__restrict__ B, cplx *__restrict__ X, double w, int nb,
                                                      const unsigned char *__restrict__ cmask) {
    const size_t total = (size_t)op.n * nb;
    const int npl = op.nplanes_total;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / nb;
        const int b = (int)(e - row * nb);
        if (cmask && !cmask[b >> 3]) continue;
        const cplx *mypc = pc + (size_t)(b / cps) * npl;
        cplx dg = {0.0, 0.0};
        const double dsg = op.conj_diag ? -1.0 : 1.0;
        for (int q = 0; q < npl; ++q) { cplx dq = op.diag[row * npl + q]; dq.y *= dsg; cfma(dg, mypc[q], dq); }
        cplx r = cdiv(B[e], dg);
        X[e] = cplx{w * r.x, w * r.y};
    }
}

// Prolongation + correction of the V-cycle, X[row][b] += sum_p P.val[p] * Xc[P.col[p]][b]: a streaming update of the fine
// multivector (read + write, 32 B per entry) with ~4 gathered coarse rows per fine row.  One lane per (row, column): the
// lanes of a wavefront read whole 1-KB coarse rows, coalesced; no LDS, no reduction.  (As a MODE_ADD launch of the
// general SpMV kernel this took 422 us per call at C2, longer than the operator itself.)
__global__ __launch_bounds__(256) void prolong_add_kernel(const int *__restrict__ ptr, const int *__restrict__ col, const double *__restrict__ val,
                                                          const cplx *__restrict__ Xc, cplx *__restrict__ X, size_t total, int nb,
                                                          const unsigned char *__restrict__ cmask) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / nb;
        const int b = (int)(e - row * nb);
        if (cmask && !cmask[b >> 3]) continue;
        const int p0 = ptr[row], p1 = ptr[row + 1];
        cplx acc = X[e];
        int p = p0;
        for (; p + 4 <= p1; p += 4) {
            cplx v[4];
            double a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = val[p + u]; v[u] = Xc[(size_t)col[p + u] * nb + b]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += a[u] * v[u].x; acc.y += a[u] * v[u].y; }
        }
        for (; p < p1; ++p) {
            const double a = val[p];
            const cplx v = Xc[(size_t)col[p] * nb + b];
            acc.x += a * v.x; acc.y += a * v.y;
        }
        X[e] = acc;
    }
}
// ---- prolongation by fine tile (wae_internal.h XferTilesDev) ------------------------------------------------------------------
// One workgroup per fine tile, walking the 8-column chunks of the batch; thread (tid >> 3, tid & 7) = (row or slot, column): the 8
// lanes of a row move one 128-byte segment.  The tile's entry list (~1 500 (slot, value) pairs) is staged in LDS once and serves
// every chunk; a chunk's slots of the coarse vector are staged in LDS; the fine rows of the NEXT chunk are requested into registers
// before the current chunk's sums run.  452 us per launch at 1M unknowns and 64 columns against 632 for prolong_add_kernel (which
// gathers every fine row's ~5 coarse rows from L2: 5 GB of gathers).
// (The restriction was built the same way -- a tile's fine rows staged in LDS, one partial sum per slot and tile, a second kernel
// summing a coarse row's partials in a fixed order -- and measured: 606 + 103 us against 540 us for the tile kernel over coarse rows;
// removed.)
__device__ __forceinline__ int next_chunk(const unsigned char *cmask, int c, int nch) { while (c < nch && cmask && !cmask[c]) ++c; return c; }

__global__ __launch_bounds__(256) void prolong_tiles_kernel(XferTilesDev T, const cplx *__restrict__ Xc, cplx *__restrict__ X, int nb,
                                                            const unsigned char *__restrict__ cmask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xt_smem[];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int col = tid & 7, rl = tid >> 3;
    const int nch = (nb + 7) >> 3;
    const int s0 = T.tptr[t], ns = T.tptr[t + 1] - s0;
    const int r0 = T.row_ptr[t], nr = T.row_ptr[t + 1] - r0;
    const int e0 = T.pptr[r0], ne = T.pptr[r0 + nr] - e0;
    cplx *const sl = (cplx *)xt_smem;                                        // [maxslots][8]
    double *const ev = (double *)(sl + (size_t)T.maxslots * 8);              // [maxent]
    int *const ep = (int *)(ev + T.maxent);                                  // [257] row pointers, local
    unsigned short *const el = (unsigned short *)(ep + 260);                 // [maxent]
    for (int i = tid; i < ne; i += 256) { ev[i] = T.pval[e0 + i]; el[i] = T.ploc[e0 + i]; }
    for (int i = tid; i <= nr; i += 256) ep[i] = T.pptr[r0 + i] - e0;
    int c = next_chunk(cmask, 0, nch);
    cplx xr[8];
    auto request = [&](int cc) {                                             // the tile's rows of chunk cc -> registers
        const int b = cc * 8 + col < nb ? cc * 8 + col : nb - 1;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int r = rl + 32 * k; xr[k] = X[(size_t)(r0 + (r < nr ? r : nr - 1)) * nb + b]; }
    };
    if (c < nch) request(c);
    while (c < nch) {
        const int b = c * 8 + col;
        const bool colok = b < nb;
        __syncthreads();                                                     // (the previous chunk's slots are no longer read; first pass: the entry lists are in place)
        for (int s = rl; s < ns; s += 32) sl[s * 8 + col] = colok ? Xc[(size_t)T.clist[s0 + s] * nb + b] : cplx{0.0, 0.0};
        cplx cur[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[k] = xr[k];
        const int cn = next_chunk(cmask, c + 1, nch);
        if (cn < nch) request(cn);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = rl + 32 * k;
            if (r < nr && colok) {
                cplx acc = cur[k];
                for (int p = ep[r]; p < ep[r + 1]; ++p) { const double a = ev[p]; const cplx v = sl[(int)el[p] * 8 + col]; acc.x += a * v.x; acc.y += a * v.y; }
                X[(size_t)(r0 + r) * nb + b] = acc;
            }
        }
        c = cn;
    }
}
static size_t xfer_lds_bytes(const XferTilesDev &T) {
    return (size_t)T.maxslots * 8 * sizeof(cplx) + (size_t)T.maxent * sizeof(double) + 260 * sizeof(int) + (size_t)T.maxent * sizeof(unsigned short) + 16;
}
void launch_prolong_tiles(const XferTilesDev &T, const cplx *Xc, cplx *X, int nb, hipStream_t st, const unsigned char *cmask) {
    if (!T.ntiles || nb <= 0) return;
    hipLaunchKernelGGL(prolong_tiles_kernel, dim3(T.ntiles), dim3(256), xfer_lds_bytes(T), st, T, Xc, X, nb, cmask);
    HIP_CHECK(hipGetLastError());
}

// compact <-> full row sets (penalty-block polish, lib.hip): out[i][b] = X[rows[i]][b];  X[rows[i]][b] += D[i][b]
__global__ __launch_bounds__(256) void gather_rows_kernel(const cplx *__restrict__ X, const int *__restrict__ rows, size_t total, int nb,
                                                          cplx *__restrict__ out) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t i = e / nb;
        out[e] = X[(size_t)rows[i] * nb + (e - i * nb)];
    }
}
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const cplx *__restrict__ D, const int *__restrict__ rows, size_t total, int nb,
                                                               cplx *__restrict__ X) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t i = e / nb;
        const size_t t = (size_t)rows[i] * nb + (e - i * nb);
        const cplx x = X[t], dlt = D[e];
        X[t] = cplx{x.x + dlt.x, x.y + dlt.y};
    }
}

static inline unsigned grid_for(size_t total, unsigned cap = 4096) {
    size_t g = (total + 255) / 256;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

void launch_prolong_add(const int *ptr, const int *col, const double *val, int64_t n, const cplx *Xc, cplx *X, int nb, hipStream_t st,
                        const unsigned char *cmask) {
    const size_t total = (size_t)n * nb;
    if (!total) return;
    hipLaunchKernelGGL(prolong_add_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, st, ptr, col, val, Xc, X, total, nb, cmask);
    HIP_CHECK(hipGetLastError());
}

void launch_gather_rows(const cplx *X, const int *rows, int64_t nrows, int nb, cplx *out, hipStream_t st) {
    const size_t total = (size_t)nrows * nb;
    if (!total) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total)), dim3(256), 0, st, X, rows, total, nb, out);
    HIP_CHECK(hipGetLastError());
}
void launch_scatter_add_rows(const cplx *D, const int *rows, int64_t nrows, int nb, cplx *X, hipStream_t st) {
    const size_t total = (size_t)nrows * nb;
    if (!total) return;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for(total)), dim3(256), 0, st, D, rows, total, nb, X);
    HIP_CHECK(hipGetLastError());
}
void launch_jacobi0(const OpDev &op, const cplx *pc, int cps, const cplx *B, cplx *X, double w, int nb, hipStream_t st, const unsigned char *cmask) {
    if (op.n <= 0) return;
    hipLaunchKernelGGL(jacobi0_kernel, dim3(grid_for((size_t)op.n * nb)), dim3(256), 0, st, op, pc, cps, B, X, w, nb, cmask);
    HIP_CHECK(hipGetLastError());
}

// per-plane input column: Y[:,0] = sum_q pc[q] plane_q X[:, plane_col[q]]   (8 lanes per row)
__global__ __launch_bounds__(256) void spmv_multi_kernel(OpDev op, const cplx *__restrict__ pc, const int *__restrict__ plane_col,
                                                         const cplx *__restrict__ X, cplx *__restrict__ Y, int nb) {
    constexpr int S = 8;
    const int tid = threadIdx.x;
    const int64_t row = (int64_t)blockIdx.x * (256 / S) + tid / S;
    const int s = tid % S;
    if (row >= op.n) return;
    cplx acc = {0.0, 0.0};
    for (int g = 0; g < op.ngroups; ++g) {
        const GroupDev G = op.g[g];
        const int p0 = G.rowptr[row], p1 = G.rowptr[row + 1];
        const int np = G.nplanes;
        for (int p = p0 + s; p < p1; p += S) {
            const int j = G.col[p];
            for (int q = 0; q < np; ++q) {
                cplx a;
                if (G.is_real) a = cplx{((const double *)G.vals)[(size_t)p * np + q], 0.0};
                else { a = ((const cplx *)G.vals)[(size_t)p * np + q]; if (G.conj_vals) a.y = -a.y; }
                const cplx m = cmul(pc[G.plane0 + q], a);
                cfma(acc, m, X[(size_t)j * nb + plane_col[G.plane0 + q]]);
            }
        }
    }
    if (op.nlong) {                                          // long rows live outside the groups' arrays: walk them here
        const int li = long_row_index(op, row);
        if (li >= 0)
            for (int p = op.long_ptr[li] + s; p < op.long_ptr[li + 1]; p += S) {
                cplx a = op.long_val[p];
                if (op.long_conj) a.y = -a.y;
                const int slot = op.long_slot[p];
                cfma(acc, cmul(pc[slot], a), X[(size_t)op.long_col[p] * nb + plane_col[slot]]);
            }
    }
    for (int off = 1; off < S; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off);
        acc.y += __shfl_xor(acc.y, off);
    }
    if (s == 0) Y[row] = acc;          // single contiguous output column
}

void launch_spmv_multi(const OpDev &op, const cplx *pc, const int *plane_col, const cplx *X, cplx *Y, int nb, hipStream_t st) {
    if (op.n <= 0) return;
    hipLaunchKernelGGL(spmv_multi_kernel, dim3((unsigned)((op.n + 31) / 32)), dim3(256), 0, st, op, pc, plane_col, X, Y, nb);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
// dense coarsest level
// ---------------------------------------------------------------------------------------------------
// A_s = sum_q pc[s][q] * op(plane_q):  op = N: plane, T: plane^T, C: conj(plane)^T (pc arrives conjugated for C)
__global__ __launch_bounds__(256) void dense_assemble_kernel(const cplx *__restrict__ planes, int nplanes, int n,
                                                             const cplx *__restrict__ pc, int op, cplx *__restrict__ A) {
    const int sys = blockIdx.y;
    const size_t nn = (size_t)n * n;
    const double sg = (op == WAE_OP_C) ? -1.0 : 1.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nn; e += (size_t)gridDim.x * 256) {
        size_t src = e;
        if (op != WAE_OP_N) { size_t i = e / n, j = e - i * n; src = j * n + i; }
        cplx acc = {0.0, 0.0};
        for (int q = 0; q < nplanes; ++q) {
            cplx a = planes[(size_t)q * nn + src];
            a.y *= sg;
            cfma(acc, pc[(size_t)sys * nplanes + q], a);
        }
        A[(size_t)sys * nn + e] = acc;
    }
}

void launch_dense_assemble(const cplx *planes, int nplanes, int n, const cplx *pc, int nsys, int op, cplx *Ainv, hipStream_t st) {
    if (n <= 0 || nsys <= 0) return;
    hipLaunchKernelGGL(dense_assemble_kernel, dim3(grid_for((size_t)n * n, 256), nsys), dim3(256), 0, st, planes, nplanes, n, pc, op, Ainv);
    HIP_CHECK(hipGetLastError());
}

// In-place Gauss-Jordan inversion with partial pivoting, one 1024-thread workgroup per system (matrix in
// global memory / L2; pivot column and row staged in LDS).  n <= 2048.
__global__ __launch_bounds__(1024) void dense_invert_kernel(cplx *__restrict__ Aall, int n, int *__restrict__ status) {
    extern __shared__ unsigned char smraw[];
    cplx *colk = (cplx *)smraw;            // n
    cplx *rowk = colk + n;                 // n
    int *perm = (int *)(rowk + n);         // n
    __shared__ double red_v[1024];
    __shared__ int red_i[1024];
    __shared__ int s_piv;
    __shared__ int s_bad;
    cplx *A = Aall + (size_t)blockIdx.x * n * n;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        // pivot search in column k, rows >= k
        double best = -1.0;
        int bi = k;
        for (int i = k + tid; i < n; i += nt) {
            cplx a = A[(size_t)i * n + k];
            double m = a.x * a.x + a.y * a.y;
            if (m > best) { best = m; bi = i; }
        }
        red_v[tid] = best; red_i[tid] = bi;
        __syncthreads();
        for (int off = nt / 2; off > 0; off >>= 1) {
            if (tid < off && red_v[tid + off] > red_v[tid]) { red_v[tid] = red_v[tid + off]; red_i[tid] = red_i[tid + off]; }
            __syncthreads();
        }
        if (tid == 0) {
            s_piv = red_i[0];
            perm[k] = red_i[0];
            if (!(red_v[0] > 0.0)) s_bad = 1;
        }
        __syncthreads();
        const int r = s_piv;
        if (r != k) {
            for (int j = tid; j < n; j += nt) {
                cplx t = A[(size_t)k * n + j];
                A[(size_t)k * n + j] = A[(size_t)r * n + j];
                A[(size_t)r * n + j] = t;
            }
        }
        __syncthreads();
        const cplx piv = A[(size_t)k * n + k];
        const cplx pinv = cdiv(cplx{1.0, 0.0}, piv);
        for (int i = tid; i < n; i += nt) colk[i] = A[(size_t)i * n + k];
        __syncthreads();
        for (int j = tid; j < n; j += nt) {
            cplx a = (j == k) ? cplx{1.0, 0.0} : A[(size_t)k * n + j];
            a = cmul(a, pinv);
            rowk[j] = a;
            A[(size_t)k * n + j] = a;
        }
        for (int i = tid; i < n; i += nt)
            if (i != k) A[(size_t)i * n + k] = cplx{0.0, 0.0};
        __syncthreads();
        const size_t nn = (size_t)n * n;
        for (size_t e = tid; e < nn; e += nt) {
            const int i = (int)(e / n), j = (int)(e - (size_t)i * n);
            if (i == k) continue;
            const cplx f = colk[i];
            cplx a = A[e];
            const cplx rk = rowk[j];
            a.x -= f.x * rk.x - f.y * rk.y;
            a.y -= f.x * rk.y + f.y * rk.x;
            A[e] = a;
        }
        __syncthreads();
    }
    // undo the row interchanges as column interchanges, in reverse order
    for (int k = n - 1; k >= 0; --k) {
        const int r = perm[k];
        if (r != k) {
            for (int i = tid; i < n; i += nt) {
                cplx t = A[(size_t)i * n + k];
                A[(size_t)i * n + k] = A[(size_t)i * n + r];
                A[(size_t)i * n + r] = t;
            }
        }
        __syncthreads();
    }
    if (tid == 0 && s_bad) atomicOr(status, 1);
}

void launch_dense_invert(cplx *Ainv, int n, int nsys, int *status, hipStream_t st) {
    if (n <= 0 || nsys <= 0) return;
    if (n > 2048) throw WaeError(WAE_ERR_INVALID, "dense coarse level too large (n > 2048)");
    size_t shm = (size_t)n * (2 * sizeof(cplx) + sizeof(int));
    hipLaunchKernelGGL(dense_invert_kernel, dim3(nsys), dim3(1024), shm, st, Ainv, n, status);
    HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void dense_apply_kernel(const cplx *__restrict__ Ainv, int n, int cps, const cplx *__restrict__ X,
                                                          cplx *__restrict__ Y, int nb, const unsigned char *__restrict__ cmask) {
    const size_t total = (size_t)n * nb;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int i = (int)(e / nb), b = (int)(e - (size_t)i * nb);
    if (cmask && !cmask[b >> 3]) return;
    const cplx *Arow = Ainv + ((size_t)(b / cps) * n + i) * n;
    cplx acc = {0.0, 0.0};
    for (int j = 0; j < n; ++j) cfma(acc, Arow[j], X[(size_t)j * nb + b]);
    Y[e] = acc;
}

void launch_dense_apply(const cplx *Ainv, int n, int cps, const cplx *X, cplx *Y, int nb, hipStream_t st, const unsigned char *cmask) {
    if (n <= 0) return;
    hipLaunchKernelGGL(dense_apply_kernel, dim3((unsigned)(((size_t)n * nb + 255) / 256)), dim3(256), 0, st, Ainv, n, cps, X, Y, nb, cmask);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
// streaming vector kernels on interleaved multivectors
// ---------------------------------------------------------------------------------------------------
__global__ void fill_zero_kernel(cplx *X, size_t count) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (size_t)gridDim.x * 256) X[e] = cplx{0.0, 0.0};
}
void launch_fill_zero(cplx *X, size_t count, hipStream_t st) {
    if (!count) return;
    hipLaunchKernelGGL(fill_zero_kernel, dim3(grid_for(count)), dim3(256), 0, st, X, count);
    HIP_CHECK(hipGetLastError());
}
void launch_copy(const cplx *X, cplx *Y, size_t count, hipStream_t st) {
    if (!count) return;
    HIP_CHECK(hipMemcpyAsync(Y, X, count * sizeof(cplx), hipMemcpyDeviceToDevice, st));
}
__global__ void add_kernel(const cplx *__restrict__ X, cplx *__restrict__ Y, size_t count) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < count; e += (size_t)gridDim.x * 256) {
        cplx a = X[e], b = Y[e];
        Y[e] = cplx{a.x + b.x, a.y + b.y};
    }
}
void launch_add(const cplx *X, cplx *Y, size_t count, hipStream_t st) {
    if (!count) return;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(count)), dim3(256), 0, st, X, Y, count);
    HIP_CHECK(hipGetLastError());
}

// partial[blk][i][b] = sum over this block's rows of conj(V_i[row][b]) W[row][b];  any nb <= 256
// (thread t owns column t % nb and every R-th row, R = 256 / nb; threads beyond R*nb idle)
constexpr int DOT_BLOCKS = 1024;
template <int MAXV, bool POW2>
__global__ __launch_bounds__(256) void dots_kernel(const cplx *__restrict__ V, size_t stride, int nv, const cplx *__restrict__ W,
                                                   int64_t n, int nb, cplx *__restrict__ partial,
                                                   const unsigned char *__restrict__ cmask) {
    __shared__ cplx sm[256];
    const int tid = threadIdx.x;
    const int R = 256 / nb;
    const int b = tid % nb, rl = tid / nb;
    const bool live = rl < R && (!cmask || cmask[b >> 3]);
    cplx acc[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) acc[i] = cplx{0.0, 0.0};
    if (live) {
        for (int64_t row = (int64_t)blockIdx.x * R + rl; row < n; row += (int64_t)gridDim.x * R) {
            const size_t e = (size_t)row * nb + b;
            const cplx w = W[e];
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                if (i < nv) {
                    const cplx v = stream_load(V + (size_t)i * stride + e);
                    acc[i].x += v.x * w.x + v.y * w.y;
                    acc[i].y += v.x * w.y - v.y * w.x;
                }
            }
        }
    }
    // block reduction over the R row-groups.  When nb is a power of two (<= 64) the lanes of a wavefront that own the
    // same column are reduced with xor shuffles first and only one value per wavefront and column goes through LDS; the
    // general case sums the R LDS entries of a column serially (at nb = 1 that was 256 serial reads per vector: 168 us
    // per launch in the narrow-batch solves of the Newton-type iterations).
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (i < nv) {
            if (POW2) {
                cplx v = acc[i];
                for (int m = 32; m >= nb; m >>= 1) {
                    v.x += __shfl_xor(v.x, m);
                    v.y += __shfl_xor(v.y, m);
                }
                const int lane = tid & 63, wv = tid >> 6;
                if (lane < nb) sm[wv * nb + lane] = v;
                __syncthreads();
                if (tid < nb) {
                    cplx s = sm[tid];
                    for (int k = 1; k < 4; ++k) { s.x += sm[k * nb + tid].x; s.y += sm[k * nb + tid].y; }
                    partial[((size_t)blockIdx.x * nv + i) * nb + tid] = s;
                }
                __syncthreads();
            } else {
                sm[tid] = acc[i];
                __syncthreads();
                if (tid < nb) {
                    cplx s = sm[tid];
                    for (int k = 1; k < R; ++k) { s.x += sm[k * nb + tid].x; s.y += sm[k * nb + tid].y; }
                    partial[((size_t)blockIdx.x * nv + i) * nb + tid] = s;
                }
                __syncthreads();
            }
        }
    }
}
// out[e] = sum_k partial[k][e]: 32 outputs x 8 k-slices per workgroup, LDS tree over the slices
// EPB outputs per workgroup, 256/EPB slices of the nblk partials each: with 32 outputs per workgroup the norms of one batch
// (64 outputs, 768-1024 partials) ran on 2 workgroups, ~100 dependent-latency loads per lane (31 us per call, 3.7 % of a pass)
template <int EPB>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const cplx *__restrict__ partial, int nblk, int count, cplx *__restrict__ out, int do_sqrt,
                                                              const cplx *__restrict__ scale, cplx *__restrict__ inv_out) {
    __shared__ cplx sm[256];
    constexpr int NS = 256 / EPB;
    const int lane_e = threadIdx.x % EPB, slice = threadIdx.x / EPB;
    const int e = blockIdx.x * EPB + lane_e;
    cplx acc = {0.0, 0.0};
    if (e < count)
        for (int k = slice; k < nblk; k += NS) { cplx p = partial[(size_t)k * count + e]; acc.x += p.x; acc.y += p.y; }
    sm[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int s = NS / 2; s >= 1; s >>= 1) {
        if (slice < s) { sm[threadIdx.x].x += sm[threadIdx.x + s * EPB].x; sm[threadIdx.x].y += sm[threadIdx.x + s * EPB].y; }
        __syncthreads();
    }
    if (slice == 0 && e < count) {
        acc = sm[threadIdx.x];
        if (scale) { const double sc = scale[e].x; acc.x *= sc; acc.y *= sc; }        // dots against unnormalised vectors (lazy GMRES basis)
        if (do_sqrt) {
            if (inv_out) inv_out[e] = cplx{acc.x > 0.0 ? 1.0 / acc.x : 0.0, 0.0};     // 1/||.||^2 of the vector just measured
            acc = cplx{sqrt(acc.x), 0.0};
        }
        out[e] = acc;
    }
}
static void launch_reduce_partials(const cplx *partial, int nblk, int count, cplx *out, int do_sqrt, hipStream_t st,
                                   const cplx *scale = nullptr, cplx *inv_out = nullptr) {
    if (count <= 256) hipLaunchKernelGGL(reduce_partials_kernel<2>, dim3((count + 1) / 2), dim3(256), 0, st, partial, nblk, count, out, do_sqrt, scale, inv_out);
    else hipLaunchKernelGGL(reduce_partials_kernel<8>, dim3((count + 7) / 8), dim3(256), 0, st, partial, nblk, count, out, do_sqrt, scale, inv_out);
    HIP_CHECK(hipGetLastError());
}

static void dots_impl(const cplx *V, size_t stride, int nv, const cplx *W, int64_t n, int nb, cplx *partial, cplx *out, int do_sqrt, hipStream_t st,
                      const unsigned char *cmask, const cplx *scale = nullptr) {
    if (nb < 1 || nb > 256) throw WaeError(WAE_ERR_INVALID, "dots: nb must be in 1..256");
    int done = 0;
    while (done < nv) {
        int chunk = nv - done > 32 ? 32 : nv - done;
        const cplx *Vc = V + (size_t)done * stride;
        // one resident round only: dots_kernel<32> holds 3 waves/SIMD (768 workgroups on 256 CUs); a 1024-block grid ran a
        // second, one-third-full round
        // small problems (narrow batches): fewer, fuller workgroups -- the second-stage reduction reads nblk partials per output
        const int64_t steps = (n + (256 / nb) - 1) / (256 / nb);
        const int nblk = (int)std::max<int64_t>(32, std::min<int64_t>(chunk <= 16 ? DOT_BLOCKS : 768, (steps + 3) / 4));
        const bool pow2 = nb <= 64 && (nb & (nb - 1)) == 0;     // wavefront-shuffle reduction needs the columns to tile a wavefront
#define WAE_DOTS(MV) do { if (pow2) hipLaunchKernelGGL((dots_kernel<MV, true>), dim3(nblk), dim3(256), 0, st, Vc, stride, chunk, W, n, nb, partial, cmask); \
                          else hipLaunchKernelGGL((dots_kernel<MV, false>), dim3(nblk), dim3(256), 0, st, Vc, stride, chunk, W, n, nb, partial, cmask); } while (0)
        if (chunk <= 8) WAE_DOTS(8);
        else if (chunk <= 16) WAE_DOTS(16);
        else WAE_DOTS(32);
#undef WAE_DOTS
        HIP_CHECK(hipGetLastError());
        int count = chunk * nb;
        launch_reduce_partials(partial, nblk, count, out + (size_t)done * nb, do_sqrt, st, scale ? scale + (size_t)done * nb : nullptr);
        done += chunk;
    }
}
// block version: partial[blk][i*nw + j][b] = sum over this block's rows of conj(V_i[row][b]) W_j[row][b], j < nw <= NW.
// Reads V once for NW right-hand vectors (the projected-operator build of the snapshot basis, lib.hip rb_append, is a
// tall-skinny Gram product: with one w per launch it re-read the whole basis for every new column).
template <int MAXV, int NW>
__global__ __launch_bounds__(256) void dots_multi_kernel(const cplx *__restrict__ V, size_t sv, int nv, const cplx *__restrict__ W, size_t sw, int nw,
                                                         int64_t n, int nb, cplx *__restrict__ partial) {
    __shared__ cplx sm[256];
    const int tid = threadIdx.x;
    const int R = 256 / nb;
    const int b = tid % nb, rl = tid / nb;
    cplx acc[MAXV][NW];
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[i][j] = cplx{0.0, 0.0};
    if (rl < R) {
        for (int64_t row = (int64_t)blockIdx.x * R + rl; row < n; row += (int64_t)gridDim.x * R) {
            const size_t e = (size_t)row * nb + b;
            cplx w[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) w[j] = j < nw ? W[(size_t)j * sw + e] : cplx{0.0, 0.0};
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                if (i < nv) {
                    const cplx v = V[(size_t)i * sv + e];
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        acc[i][j].x += v.x * w[j].x + v.y * w[j].y;
                        acc[i][j].y += v.x * w[j].y - v.y * w[j].x;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            if (i < nv && j < nw) {
                sm[tid] = acc[i][j];
                __syncthreads();
                if (tid < nb) {
                    cplx s = sm[tid];
                    for (int k = 1; k < R; ++k) { s.x += sm[k * nb + tid].x; s.y += sm[k * nb + tid].y; }
                    partial[((size_t)blockIdx.x * nv * nw + (size_t)i * nw + j) * nb + tid] = s;
                }
                __syncthreads();
            }
        }
    }
}
// out[(i*nw + j)*nb + b] = V_i[:,b]^H W_j[:,b]   (nv arbitrary, nw <= 4)
void launch_dots_multi(const cplx *V, size_t sv, int nv, const cplx *W, size_t sw, int nw, int64_t n, int nb, cplx *partial, cplx *out,
                       hipStream_t st) {
    if (nb < 1 || nb > 256 || nw < 1 || nw > 4) throw WaeError(WAE_ERR_INVALID, "dots_multi: nb in 1..256, nw in 1..4");
    int done = 0;
    while (done < nv) {
        const int chunk = std::min(8, nv - done);
        const int nblk = 768;
        hipLaunchKernelGGL((dots_multi_kernel<8, 4>), dim3(nblk), dim3(256), 0, st, V + (size_t)done * sv, sv, chunk, W, sw, nw, n, nb, partial);
        HIP_CHECK(hipGetLastError());
        const int count = chunk * nw * nb;
        launch_reduce_partials(partial, nblk, count, out + (size_t)done * nw * nb, 0, st);
        done += chunk;
    }
}
void launch_dots(const cplx *V, size_t stride, int nv, const cplx *W, int64_t n, int nb, cplx *partial, cplx *out, hipStream_t st,
                 const unsigned char *cmask) {
    dots_impl(V, stride, nv, W, n, nb, partial, out, 0, st, cmask);
}
void launch_dots_scaled(const cplx *V, size_t stride, int nv, const cplx *W, int64_t n, int nb, cplx *partial, cplx *out, const cplx *scale,
                        hipStream_t st, const unsigned char *cmask) {
    dots_impl(V, stride, nv, W, n, nb, partial, out, 0, st, cmask, scale);
}
void launch_norms(const cplx *X, int64_t n, int nb, cplx *partial, cplx *out, hipStream_t st, const unsigned char *cmask) {
    dots_impl(X, 0, 1, X, n, nb, partial, out, 1, st, cmask);
}

// W[row][b] = base[row][b] + sign * sum_i c[i][b] V_i[row][b].  The nv x nb coefficients are staged in LDS once per
// workgroup (reading them per element through the vector cache doubled the L1 traffic of this streaming kernel), and
// the basis vectors are fetched AXU at a time so that AXU 16-B loads are in flight per lane.
// Thread t owns column t % nb and every R-th row, R = 256 / nb (as in dots_kernel).
constexpr int AXU = 8;
constexpr int AX_MAXC = 4096;      // coefficients per launch (64 KB of LDS)
// NORM: additionally partial[blk][b] = sum over this workgroup's rows of |W[row][b]|^2 (the Gram-Schmidt step needs the norm
// of the vector it has just written: one pass over it less)
template <bool NORM>
__global__ __launch_bounds__(256) void axpy_neg_kernel(const cplx *__restrict__ V, size_t stride, int nv, const cplx *__restrict__ h,
                                                       cplx *W, int64_t n, int nb, double sign, const cplx *base,
                                                       const unsigned char *__restrict__ cmask, cplx *__restrict__ partial) {
    extern __shared__ cplx hs[];
    const int tid = threadIdx.x;
    for (int k = tid; k < nv * nb; k += 256) {
        const cplx c = h[k];
        hs[k] = cplx{sign * c.x, sign * c.y};
    }
    __syncthreads();
    const int R = 256 / nb;
    const int b = tid % nb, rl = tid / nb;
    const bool live = rl < R && !(cmask && !cmask[b >> 3]);
    if (!NORM && !live) return;
    double nrm2 = 0.0;
    if (live)
    for (int64_t row = (int64_t)blockIdx.x * R + rl; row < n; row += (int64_t)gridDim.x * R) {
        const size_t e = (size_t)row * nb + b;
        cplx acc = base ? base[e] : cplx{0.0, 0.0};
        int i = 0;
        for (; i + AXU <= nv; i += AXU) {
            cplx v[AXU];
#pragma unroll
            for (int u = 0; u < AXU; ++u) v[u] = stream_load(V + (size_t)(i + u) * stride + e);
#pragma unroll
            for (int u = 0; u < AXU; ++u) {
                const cplx c = hs[(i + u) * nb + b];
                acc.x += c.x * v[u].x - c.y * v[u].y;
                acc.y += c.x * v[u].y + c.y * v[u].x;
            }
        }
        for (; i < nv; ++i) {
            const cplx c = hs[i * nb + b];
            const cplx v = stream_load(V + (size_t)i * stride + e);
            acc.x += c.x * v.x - c.y * v.y;
            acc.y += c.x * v.y + c.y * v.x;
        }
        W[e] = acc;
        if (NORM) nrm2 += acc.x * acc.x + acc.y * acc.y;
    }
    if (NORM) {
        __syncthreads();                        // hs is re-used for the reduction
        double *sm = (double *)hs;
        if (nb <= 64 && (nb & (nb - 1)) == 0) {
            for (int m = 32; m >= nb; m >>= 1) nrm2 += __shfl_xor(nrm2, m);
            const int lane = tid & 63, wv = tid >> 6;
            if (lane < nb) sm[wv * nb + lane] = nrm2;
            __syncthreads();
            if (tid < nb) partial[(size_t)blockIdx.x * nb + tid] = cplx{sm[tid] + sm[nb + tid] + sm[2 * nb + tid] + sm[3 * nb + tid], 0.0};
        } else {
            sm[tid] = nrm2;
            __syncthreads();
            if (tid < nb) {
                double sacc = sm[tid];
                for (int k = 1; k < R; ++k) sacc += sm[k * nb + tid];
                partial[(size_t)blockIdx.x * nb + tid] = cplx{sacc, 0.0};
            }
        }
    }
}
static void axpy_impl(const cplx *V, size_t stride, int nv, const cplx *c, cplx *W, int64_t n, int nb, double sign, const cplx *base,
                      hipStream_t st, const unsigned char *cmask) {
    if (!n || nb < 1) return;
    if (nb > 256) throw WaeError(WAE_ERR_INVALID, "axpy: nb must be in 1..256");
    const int R = 256 / nb;
    const int64_t steps = (n + R - 1) / R;
    const unsigned grid = (unsigned)std::min<int64_t>(steps, 2048);
    const int maxv = std::max(1, AX_MAXC / nb);
    int done = 0;
    do {                                              // nv == 0 still writes W = base (or 0)
        const int chunk = std::min(nv - done, maxv);
        hipLaunchKernelGGL(axpy_neg_kernel<false>, dim3(grid), dim3(256), (size_t)std::max(chunk, 1) * nb * sizeof(cplx), st,
                           V + (size_t)done * stride, stride, chunk, c + (size_t)done * nb, W, n, nb, sign, done ? (const cplx *)W : base, cmask,
                           (cplx *)nullptr);
        HIP_CHECK(hipGetLastError());
        done += chunk;
    } while (done < nv);
}
void launch_axpy_neg(const cplx *V, size_t stride, int nv, const cplx *h, cplx *W, int64_t n, int nb, hipStream_t st, const unsigned char *cmask) {
    axpy_impl(V, stride, nv, h, W, n, nb, -1.0, W, st, cmask);
}
// W_j -= sum_i h[i][j] V_i for CNT vectors W_j (stride wstride) in ONE reading of V_0..nv-1: the block Gram-Schmidt update of the
// snapshot basis (lib.hip rb_append_block), whose four new vectors used to read the basis once each.  h[(i*CNT + j)*nb + b], the
// layout dots_multi writes.
template <int CNT>
__global__ __launch_bounds__(256) void axpy_neg_multi_kernel(const cplx *__restrict__ V, size_t stride, int nv, const cplx *__restrict__ h, cplx *W,
                                                             size_t wstride, int64_t n, int nb) {
    extern __shared__ cplx hs[];
    const int tid = threadIdx.x;
    for (int k = tid; k < nv * CNT * nb; k += 256) {
        const cplx c = h[k];
        hs[k] = cplx{-c.x, -c.y};
    }
    __syncthreads();
    const int R = 256 / nb;
    const int b = tid % nb, rl = tid / nb;
    if (rl >= R) return;
    for (int64_t row = (int64_t)blockIdx.x * R + rl; row < n; row += (int64_t)gridDim.x * R) {
        const size_t e = (size_t)row * nb + b;
        cplx acc[CNT];
#pragma unroll
        for (int j = 0; j < CNT; ++j) acc[j] = W[(size_t)j * wstride + e];
        for (int i = 0; i < nv; ++i) {
            const cplx v = stream_load(V + (size_t)i * stride + e);
#pragma unroll
            for (int j = 0; j < CNT; ++j) {
                const cplx c = hs[(i * CNT + j) * nb + b];
                acc[j].x += c.x * v.x - c.y * v.y;
                acc[j].y += c.x * v.y + c.y * v.x;
            }
        }
#pragma unroll
        for (int j = 0; j < CNT; ++j) W[(size_t)j * wstride + e] = acc[j];
    }
}
void launch_axpy_neg_multi(const cplx *V, size_t stride, int nv, const cplx *h, cplx *W, size_t wstride, int cnt, int64_t n, int nb, hipStream_t st) {
    if (!n || nb < 1 || nv < 1 || cnt < 1) return;
    const size_t shm = (size_t)nv * cnt * nb * sizeof(cplx);
    if (nb > 256 || cnt > 4 || shm > 60 * 1024) throw WaeError(WAE_ERR_INVALID, "axpy_neg_multi: coefficients do not fit one launch");
    const int R = 256 / nb;
    const int64_t steps = (n + R - 1) / R;
    const unsigned grid = (unsigned)std::min<int64_t>(steps, 2048);
    switch (cnt) {
    case 1: hipLaunchKernelGGL(axpy_neg_multi_kernel<1>, dim3(grid), dim3(256), shm, st, V, stride, nv, h, W, wstride, n, nb); break;
    case 2: hipLaunchKernelGGL(axpy_neg_multi_kernel<2>, dim3(grid), dim3(256), shm, st, V, stride, nv, h, W, wstride, n, nb); break;
    case 3: hipLaunchKernelGGL(axpy_neg_multi_kernel<3>, dim3(grid), dim3(256), shm, st, V, stride, nv, h, W, wstride, n, nb); break;
    default: hipLaunchKernelGGL(axpy_neg_multi_kernel<4>, dim3(grid), dim3(256), shm, st, V, stride, nv, h, W, wstride, n, nb); break;
    }
    HIP_CHECK(hipGetLastError());
}
// w -= V h and norms[b] = ||w[:,b]|| in one pass over w (falls back to two kernels when the coefficients do not fit one launch)
// base (optional): W = base - V h instead of the in-place update; inv_out (optional): 1/||W[:,b]||^2 beside the norms.  Both are
// what a Krylov basis kept UNNORMALISED needs (lib.hip gmres): the new vector goes straight into its basis slot.
void launch_axpy_neg_norm(const cplx *V, size_t stride, int nv, const cplx *h, cplx *W, int64_t n, int nb, cplx *partial, cplx *norms,
                          hipStream_t st, const unsigned char *cmask, const cplx *base, cplx *inv_out) {
    if (!n || nb < 1) return;
    if (nb > 256) throw WaeError(WAE_ERR_INVALID, "axpy: nb must be in 1..256");
    if (!base) base = W;
    if (nv < 1 || nv > AX_MAXC / nb) {
        if (inv_out) throw WaeError(WAE_ERR_INVALID, "axpy_neg_norm: inverse norms need the single-launch form");
        axpy_impl(V, stride, nv, h, W, n, nb, -1.0, base, st, cmask);
        launch_norms(W, n, nb, partial, norms, st, cmask);
        return;
    }
    const int R = 256 / nb;
    const int64_t steps = (n + R - 1) / R;
    const unsigned grid = (unsigned)std::min<int64_t>(steps, 1024);
    const size_t shm = std::max((size_t)nv * nb * sizeof(cplx), (size_t)256 * sizeof(double));
    hipLaunchKernelGGL(axpy_neg_kernel<true>, dim3(grid), dim3(256), shm, st, V, stride, nv, h, W, n, nb, -1.0, base, cmask, partial);
    HIP_CHECK(hipGetLastError());
    launch_reduce_partials(partial, (int)grid, nb, norms, 1, st, nullptr, inv_out);
}
// ---------------------------------------------------------------------------------------------------
// Two Arnoldi steps per pass over the basis (lib.hip gmres_wide, "pair" steps).  With w1 = Op v_j and w2 = Op w1 -- the operator
// applied to the vector BEFORE it is orthogonalised -- both new basis vectors come out of ONE reading of V_0..j for the inner
// products and ONE for the update, where two single steps read it four times: the Gram-Schmidt traffic of a long recurrence, the
// largest stream of a from-zero solve, halves.
//   dots2:  c1 = V^H w1, c2 = V^H w2 (each scaled by 1/||v_i||^2: coefficients against the unnormalised basis) and the Gram
//           entries w1^H w1, w1^H w2, w2^H w2;
//   axpy2:  v_{j+1} = w1 - V c1,   v_{j+2} = w2 - alpha w1 - V (c2 - alpha c1)   in place, with their squared norms.
// ---------------------------------------------------------------------------------------------------
template <int MAXV, bool POW2>
__global__ __launch_bounds__(256) void dots2_kernel(const cplx *__restrict__ V, size_t stride, int nv, const cplx *__restrict__ W1,
                                                    const cplx *__restrict__ W2, int64_t n, int nb, cplx *__restrict__ partial,
                                                    const unsigned char *__restrict__ cmask, int gram) {
    __shared__ cplx sm[256];
    const int tid = threadIdx.x;
    const int R = 256 / nb;
    const int b = tid % nb, rl = tid / nb;
    const bool live = rl < R && (!cmask || cmask[b >> 3]);
    cplx a1[MAXV], a2[MAXV];
    cplx g[3];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { a1[i] = cplx{0.0, 0.0}; a2[i] = cplx{0.0, 0.0}; }
    g[0] = g[1] = g[2] = cplx{0.0, 0.0};
    if (live) {
        for (int64_t row = (int64_t)blockIdx.x * R + rl; row < n; row += (int64_t)gridDim.x * R) {
            const size_t e = (size_t)row * nb + b;
            const cplx w1 = W1[e], w2 = W2[e];
            if (gram) {
                g[0].x += w1.x * w1.x + w1.y * w1.y;
                g[1].x += w1.x * w2.x + w1.y * w2.y; g[1].y += w1.x * w2.y - w1.y * w2.x;
                g[2].x += w2.x * w2.x + w2.y * w2.y;
            }
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                if (i < nv) {
                    const cplx v = stream_load(V + (size_t)i * stride + e);
                    a1[i].x += v.x * w1.x + v.y * w1.y; a1[i].y += v.x * w1.y - v.y * w1.x;
                    a2[i].x += v.x * w2.x + v.y * w2.y; a2[i].y += v.x * w2.y - v.y * w2.x;
                }
            }
        }
    }
    // block reduction (as dots_kernel); output order [k][i][b], k = 0, 1, then the three Gram entries
    const int nout = 2 * nv + (gram ? 3 : 0);
    auto reduce_store = [&](cplx v, int slot) {
        if (POW2) {
            for (int m = 32; m >= nb; m >>= 1) { v.x += __shfl_xor(v.x, m); v.y += __shfl_xor(v.y, m); }
            const int lane = tid & 63, wv = tid >> 6;
            if (lane < nb) sm[wv * nb + lane] = v;
            __syncthreads();
            if (tid < nb) {
                cplx s = sm[tid];
                for (int k = 1; k < 4; ++k) { s.x += sm[k * nb + tid].x; s.y += sm[k * nb + tid].y; }
                partial[((size_t)blockIdx.x * nout + slot) * nb + tid] = s;
            }
            __syncthreads();
        } else {
            sm[tid] = v;
            __syncthreads();
            if (tid < nb) {
                cplx s = sm[tid];
                for (int k = 1; k < R; ++k) { s.x += sm[k * nb + tid].x; s.y += sm[k * nb + tid].y; }
                partial[((size_t)blockIdx.x * nout + slot) * nb + tid] = s;
            }
            __syncthreads();
        }
    };
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv) { reduce_store(a1[i], i); reduce_store(a2[i], nv + i); }
    if (gram)
        for (int q = 0; q < 3; ++q) reduce_store(g[q], 2 * nv + q);
}
// second stage: out1[i][b], out2[i][b] (scaled by scale[i][b].x) and gram[q][b] from partial[blk][2 nv + 3][nb]
__global__ __launch_bounds__(256) void reduce_partials2_kernel(const cplx *__restrict__ partial, int nblk, int nv, int nb, int gram, cplx *__restrict__ out1,
                                                               cplx *__restrict__ out2, const cplx *__restrict__ scale, cplx *__restrict__ gram_out) {
    __shared__ cplx sm[256];
    constexpr int EPB = 2, NS = 256 / EPB;
    const int count = (2 * nv + (gram ? 3 : 0)) * nb;
    const int lane_e = threadIdx.x % EPB, slice = threadIdx.x / EPB;
    const int e = blockIdx.x * EPB + lane_e;
    cplx acc = {0.0, 0.0};
    if (e < count)
        for (int k = slice; k < nblk; k += NS) { const cplx p = partial[(size_t)k * count + e]; acc.x += p.x; acc.y += p.y; }
    sm[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int s = NS / 2; s >= 1; s >>= 1) {
        if (slice < s) { sm[threadIdx.x].x += sm[threadIdx.x + s * EPB].x; sm[threadIdx.x].y += sm[threadIdx.x + s * EPB].y; }
        __syncthreads();
    }
    if (slice == 0 && e < count) {
        acc = sm[threadIdx.x];
        const int slot = e / nb, b = e - slot * nb;
        if (slot < 2 * nv) {
            const int i = slot < nv ? slot : slot - nv;
            const double sc = scale[(size_t)i * nb + b].x;
            (slot < nv ? out1 : out2)[(size_t)i * nb + b] = cplx{acc.x * sc, acc.y * sc};
        } else {
            gram_out[(size_t)(slot - 2 * nv) * nb + b] = acc;
        }
    }
}
void launch_dots2_scaled(const cplx *V, size_t stride, int nv, const cplx *W1, const cplx *W2, int64_t n, int nb, cplx *partial, cplx *out1,
                         cplx *out2, cplx *gram_out, const cplx *scale, hipStream_t st, const unsigned char *cmask) {
    if (nb < 1 || nb > 256) throw WaeError(WAE_ERR_INVALID, "dots2: nb must be in 1..256");
    const bool pow2 = nb <= 64 && (nb & (nb - 1)) == 0;
    const int64_t steps = (n + (256 / nb) - 1) / (256 / nb);
    const int nblk = (int)std::max<int64_t>(32, std::min<int64_t>(768, (steps + 3) / 4));
    int done = 0;
    do {
        const int chunk = std::min(16, nv - done);
        const int gram = done == 0 ? 1 : 0;
        const cplx *Vc = V + (size_t)done * stride;
        if (pow2) hipLaunchKernelGGL((dots2_kernel<16, true>), dim3(nblk), dim3(256), 0, st, Vc, stride, chunk, W1, W2, n, nb, partial, cmask, gram);
        else hipLaunchKernelGGL((dots2_kernel<16, false>), dim3(nblk), dim3(256), 0, st, Vc, stride, chunk, W1, W2, n, nb, partial, cmask, gram);
        HIP_CHECK(hipGetLastError());
        const int count = (2 * chunk + (gram ? 3 : 0)) * nb;
        hipLaunchKernelGGL(reduce_partials2_kernel, dim3((count + 1) / 2), dim3(256), 0, st, partial, nblk, chunk, nb, gram, out1 + (size_t)done * nb,
                           out2 + (size_t)done * nb, scale + (size_t)done * nb, gram_out);
        HIP_CHECK(hipGetLastError());
        done += chunk;
    } while (done < nv);
}

// v1 = w1 - V c1,  v2 = w2 - alpha w1 - V c2m  (c2m = c2 - alpha c1), in place of w1 / w2; partial[blk][k][b] = this workgroup's part
// of ||v_k||^2.  Coefficients staged in LDS ([2][nv][nb]); 512 threads so that one workgroup per CU keeps ~64 KB of loads in flight.
template <int NT>
__global__ __launch_bounds__(NT) void axpy2_kernel(const cplx *__restrict__ V, size_t stride, int nv, const cplx *__restrict__ c1, const cplx *__restrict__ c2m,
                                                   const cplx *__restrict__ alpha, cplx *W1, cplx *W2, int64_t n, int nb,
                                                   const unsigned char *__restrict__ cmask, cplx *__restrict__ partial) {
    extern __shared__ cplx hs[];
    const int tid = threadIdx.x;
    for (int k = tid; k < nv * nb; k += NT) { hs[k] = c1[k]; hs[nv * nb + k] = c2m[k]; }
    __syncthreads();
    const int R = NT / nb;
    const int b = tid % nb, rl = tid / nb;
    const bool live = rl < R && !(cmask && !cmask[b >> 3]);
    double n1 = 0.0, n2 = 0.0;
    if (live) {
        const cplx al = alpha[b];
        const cplx *h1 = hs + b, *h2 = hs + (size_t)nv * nb + b;
        for (int64_t row = (int64_t)blockIdx.x * R + rl; row < n; row += (int64_t)gridDim.x * R) {
            const size_t e = (size_t)row * nb + b;
            const cplx w1 = W1[e], w2 = W2[e];
            cplx o1 = w1;
            cplx o2 = {w2.x - (al.x * w1.x - al.y * w1.y), w2.y - (al.x * w1.y + al.y * w1.x)};
            int i = 0;
            for (; i + AXU <= nv; i += AXU) {
                cplx v[AXU];
#pragma unroll
                for (int u = 0; u < AXU; ++u) v[u] = stream_load(V + (size_t)(i + u) * stride + e);
#pragma unroll
                for (int u = 0; u < AXU; ++u) {
                    const cplx p = h1[(size_t)(i + u) * nb], q = h2[(size_t)(i + u) * nb];
                    o1.x -= p.x * v[u].x - p.y * v[u].y; o1.y -= p.x * v[u].y + p.y * v[u].x;
                    o2.x -= q.x * v[u].x - q.y * v[u].y; o2.y -= q.x * v[u].y + q.y * v[u].x;
                }
            }
            for (; i < nv; ++i) {
                const cplx p = h1[(size_t)i * nb], q = h2[(size_t)i * nb];
                const cplx v = stream_load(V + (size_t)i * stride + e);
                o1.x -= p.x * v.x - p.y * v.y; o1.y -= p.x * v.y + p.y * v.x;
                o2.x -= q.x * v.x - q.y * v.y; o2.y -= q.x * v.y + q.y * v.x;
            }
            W1[e] = o1;
            W2[e] = o2;
            n1 += o1.x * o1.x + o1.y * o1.y;
            n2 += o2.x * o2.x + o2.y * o2.y;
        }
    }
    __syncthreads();                            // hs is re-used for the reduction
    double *sm = (double *)hs;
    sm[tid] = n1; sm[NT + tid] = n2;
    __syncthreads();
    if (tid < nb) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < R; ++k) { s1 += sm[k * nb + tid]; s2 += sm[NT + k * nb + tid]; }
        partial[((size_t)blockIdx.x * 2 + 0) * nb + tid] = cplx{s1, 0.0};
        partial[((size_t)blockIdx.x * 2 + 1) * nb + tid] = cplx{s2, 0.0};
    }
}
// norms[k][b] = ||v_k[:,b]|| and inv[k][b] = 1/||v_k||^2, k = 0, 1 (norms, inv: 2 x nb each)
void launch_axpy2_norm(const cplx *V, size_t stride, int nv, const cplx *c1, const cplx *c2m, const cplx *alpha, cplx *W1, cplx *W2, int64_t n, int nb,
                       cplx *partial, cplx *norms, cplx *inv_out, hipStream_t st, const unsigned char *cmask) {
    if (!n || nb < 1) return;
    if (nb > 256 || nv < 1 || (size_t)2 * nv * nb * sizeof(cplx) > 150 * 1024) throw WaeError(WAE_ERR_INVALID, "axpy2: coefficients do not fit LDS");
    static std::mutex mu;
    static bool attr_done[64] = {false};
    int dev_now = 0;
    HIP_CHECK(hipGetDevice(&dev_now));
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_done[dev_now & 63]) {
            HIP_CHECK(hipFuncSetAttribute((const void *)axpy2_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            HIP_CHECK(hipFuncSetAttribute((const void *)axpy2_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_done[dev_now & 63] = true;
        }
    }
    const size_t shm = std::max((size_t)2 * nv * nb * sizeof(cplx), (size_t)2 * 512 * sizeof(double));
    const bool big = shm > 40 * 1024;              // few workgroups fit a CU: make them large
    const int NT = big ? 512 : 256;
    const int R = NT / nb;
    const int64_t steps = (n + R - 1) / R;
    const unsigned grid = (unsigned)std::min<int64_t>(steps, big ? 512 : 1024);
    if (big) hipLaunchKernelGGL(axpy2_kernel<512>, dim3(grid), dim3(512), shm, st, V, stride, nv, c1, c2m, alpha, W1, W2, n, nb, cmask, partial);
    else hipLaunchKernelGGL(axpy2_kernel<256>, dim3(grid), dim3(256), shm, st, V, stride, nv, c1, c2m, alpha, W1, W2, n, nb, cmask, partial);
    HIP_CHECK(hipGetLastError());
    launch_reduce_partials(partial, (int)grid, 2 * nb, norms, 1, st, nullptr, inv_out);
}

void launch_lincomb(const cplx *V, size_t stride, int nv, const cplx *y, cplx *Y, int64_t n, int nb, hipStream_t st) {
    axpy_impl(V, stride, nv, y, Y, n, nb, 1.0, nullptr, st, nullptr);
}
void launch_lincomb_add(const cplx *V, size_t stride, int nv, const cplx *y, cplx *X, int64_t n, int nb, hipStream_t st) {
    axpy_impl(V, stride, nv, y, X, n, nb, 1.0, X, st, nullptr);      // X += sum_i y_i V_i
}

// ---------------------------------------------------------------------------------------------------
// snapshot-basis helpers (Galerkin initial guesses for the shifted systems of a contour, lib.hip: beyn_moments_rb)
// ---------------------------------------------------------------------------------------------------
// out[row][c] = X[row][off + c], c < l   (one system's l columns out of a lock-step batch of nb columns)
__global__ __launch_bounds__(256) void extract_cols_kernel(const cplx *__restrict__ X, int nb, int off, int l, cplx *__restrict__ out, size_t total) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / l;
        const int c = (int)(e - row * l);
        out[e] = X[row * nb + off + c];
    }
}
void launch_extract_cols(const cplx *X, int nb, int off, int l, cplx *out, int64_t n, hipStream_t st) {
    const size_t total = (size_t)n * l;
    if (!total) return;
    hipLaunchKernelGGL(extract_cols_kernel, dim3(grid_for(total)), dim3(256), 0, st, X, nb, off, l, out, total);
    HIP_CHECK(hipGetLastError());
}
// X[row][b] = sum_i y[i][b] Q_i[row][b % l]: the basis multivectors have l columns (one per probe column), the batch
// has nb = nsys*l columns (every system re-uses the same l bases with its own coefficients).  Coefficients in LDS.
__global__ __launch_bounds__(256) void lincomb_rep_kernel(const cplx *__restrict__ Q, size_t stride, int nv, const cplx *__restrict__ y,
                                                          cplx *__restrict__ X, int64_t n, int nb, int l, int accumulate) {
    extern __shared__ cplx hs[];
    const int tid = threadIdx.x;
    for (int k = tid; k < nv * nb; k += 256) hs[k] = y[k];
    __syncthreads();
    const int R = 256 / nb;
    const int b = tid % nb, rl = tid / nb;
    if (rl >= R) return;
    const int c = b % l;
    for (int64_t row = (int64_t)blockIdx.x * R + rl; row < n; row += (int64_t)gridDim.x * R) {
        const size_t eq = (size_t)row * l + c;
        cplx acc = accumulate ? X[(size_t)row * nb + b] : cplx{0.0, 0.0};
        int i = 0;
        for (; i + AXU <= nv; i += AXU) {
            cplx v[AXU];
#pragma unroll
            for (int u = 0; u < AXU; ++u) v[u] = Q[(size_t)(i + u) * stride + eq];
#pragma unroll
            for (int u = 0; u < AXU; ++u) {
                const cplx cf = hs[(i + u) * nb + b];
                acc.x += cf.x * v[u].x - cf.y * v[u].y;
                acc.y += cf.x * v[u].y + cf.y * v[u].x;
            }
        }
        for (; i < nv; ++i) {
            const cplx cf = hs[i * nb + b];
            const cplx v = Q[(size_t)i * stride + eq];
            acc.x += cf.x * v.x - cf.y * v.y;
            acc.y += cf.x * v.y + cf.y * v.x;
        }
        X[(size_t)row * nb + b] = acc;
    }
}
// The same for batches of at most 8 systems (l >= 8 probe columns at the default width): one thread per (row, probe column)
// loads every basis entry ONCE and feeds the accumulators of all systems -- in the kernel above the lanes of the nsys systems
// load the same 16 bytes each (8 x the L1 requests: 2.55 TB/s of unique reads at 1M DoF).
constexpr int LRS = 8;
__global__ __launch_bounds__(256) void lincomb_rep8_kernel(const cplx *__restrict__ Q, size_t stride, int nv, const cplx *__restrict__ y,
                                                           cplx *__restrict__ X, int64_t n, int nb, int l, int nsys, int accumulate) {
    extern __shared__ cplx hs[];
    const int tid = threadIdx.x;
    for (int k = tid; k < nv * nb; k += 256) hs[k] = y[k];
    __syncthreads();
    const int R = 256 / l;
    const int c = tid % l, rl = tid / l;
    if (rl >= R) return;
    // TWO rows per thread and step (round 4): every coefficient read from LDS serves both, and eight basis entries are in flight per
    // thread instead of four (1 214 -> ~900 us for the 40 x 8 x 1M basis of the benchmark: the kernel was bound by its LDS reads, one
    // 16-byte coefficient per basis entry and system)
    for (int64_t row0 = ((int64_t)blockIdx.x * R + rl) * 2; row0 < n; row0 += (int64_t)gridDim.x * R * 2) {
        const bool two = row0 + 1 < n;
        const size_t eq0 = (size_t)row0 * l + c, eq1 = two ? eq0 + l : eq0;
        cplx acc[2][LRS];
#pragma unroll
        for (int s = 0; s < LRS; ++s) {
            acc[0][s] = (accumulate && s < nsys) ? X[(size_t)row0 * nb + s * l + c] : cplx{0.0, 0.0};
            acc[1][s] = (accumulate && s < nsys && two) ? X[(size_t)(row0 + 1) * nb + s * l + c] : cplx{0.0, 0.0};
        }
        int i = 0;
        for (; i + 4 <= nv; i += 4) {
            cplx v0[4], v1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { v0[u] = Q[(size_t)(i + u) * stride + eq0]; v1[u] = Q[(size_t)(i + u) * stride + eq1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int s = 0; s < LRS; ++s)
                    if (s < nsys) {
                        const cplx cf = hs[(i + u) * nb + s * l + c];
                        acc[0][s].x += cf.x * v0[u].x - cf.y * v0[u].y;
                        acc[0][s].y += cf.x * v0[u].y + cf.y * v0[u].x;
                        acc[1][s].x += cf.x * v1[u].x - cf.y * v1[u].y;
                        acc[1][s].y += cf.x * v1[u].y + cf.y * v1[u].x;
                    }
        }
        for (; i < nv; ++i) {
            const cplx v0 = Q[(size_t)i * stride + eq0], v1 = Q[(size_t)i * stride + eq1];
#pragma unroll
            for (int s = 0; s < LRS; ++s)
                if (s < nsys) {
                    const cplx cf = hs[i * nb + s * l + c];
                    acc[0][s].x += cf.x * v0.x - cf.y * v0.y;
                    acc[0][s].y += cf.x * v0.y + cf.y * v0.x;
                    acc[1][s].x += cf.x * v1.x - cf.y * v1.y;
                    acc[1][s].y += cf.x * v1.y + cf.y * v1.x;
                }
        }
#pragma unroll
        for (int s = 0; s < LRS; ++s)
            if (s < nsys) {
                X[(size_t)row0 * nb + s * l + c] = acc[0][s];
                if (two) X[(size_t)(row0 + 1) * nb + s * l + c] = acc[1][s];
            }
    }
}
void launch_lincomb_rep(const cplx *Q, size_t stride, int nv, const cplx *y, cplx *X, int64_t n, int nb, int l, hipStream_t st) {
    if (!n || nb < 1) return;
    if (nb > 256) throw WaeError(WAE_ERR_INVALID, "lincomb_rep: nb must be in 1..256");
    if (l >= 4 && l <= 256 && nb % l == 0 && nb / l <= LRS) {
        const int R = 256 / l, nsys = nb / l;
        const unsigned grid = (unsigned)std::min<int64_t>((n + 2 * R - 1) / (2 * R), 4096);
        const int maxv = std::max(1, AX_MAXC / nb);
        int done = 0;
        do {
            const int chunk = std::min(nv - done, maxv);
            hipLaunchKernelGGL(lincomb_rep8_kernel, dim3(grid), dim3(256), (size_t)std::max(chunk, 1) * nb * sizeof(cplx), st,
                               Q + (size_t)done * stride, stride, chunk, y + (size_t)done * nb, X, n, nb, l, nsys, done ? 1 : 0);
            HIP_CHECK(hipGetLastError());
            done += chunk;
        } while (done < nv);
        return;
    }
    const int R = 256 / nb;
    const unsigned grid = (unsigned)std::min<int64_t>((n + R - 1) / R, 2048);
    const int maxv = std::max(1, AX_MAXC / nb);
    int done = 0;
    do {
        const int chunk = std::min(nv - done, maxv);
        hipLaunchKernelGGL(lincomb_rep_kernel, dim3(grid), dim3(256), (size_t)std::max(chunk, 1) * nb * sizeof(cplx), st,
                           Q + (size_t)done * stride, stride, chunk, y + (size_t)done * nb, X, n, nb, l, done ? 1 : 0);
        HIP_CHECK(hipGetLastError());
        done += chunk;
    } while (done < nv);
}
// X[row][b] *= keep[b] (keep = 0 or 1, real part of a complex table): drop the guesses of selected columns
__global__ __launch_bounds__(256) void mask_cols_kernel(cplx *__restrict__ X, const cplx *__restrict__ keep, size_t total, int nb) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        if (keep[e % nb].x == 0.0) X[e] = cplx{0.0, 0.0};
    }
}
void launch_mask_cols(cplx *X, const cplx *keep, int64_t n, int nb, hipStream_t st) {
    const size_t total = (size_t)n * nb;
    if (!total) return;
    hipLaunchKernelGGL(mask_cols_kernel, dim3(grid_for(total)), dim3(256), 0, st, X, keep, total, nb);
    HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void scale_inv_kernel(const cplx *__restrict__ X, const cplx *__restrict__ alpha, cplx *__restrict__ Y, size_t total, int nb,
                                                        const unsigned char *__restrict__ cmask) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        if (cmask && !cmask[(e % nb) >> 3]) continue;
        const double a = alpha[e % nb].x;
        const double s = (a > 1e-300) ? 1.0 / a : 0.0;
        const cplx x = X[e];
        Y[e] = cplx{x.x * s, x.y * s};
    }
}
void launch_scale_inv(const cplx *X, const cplx *alpha, cplx *Y, int64_t n, int nb, hipStream_t st, const unsigned char *cmask) {
    size_t total = (size_t)n * nb;
    if (!total) return;
    hipLaunchKernelGGL(scale_inv_kernel, dim3(grid_for(total)), dim3(256), 0, st, X, alpha, Y, total, nb, cmask);
    HIP_CHECK(hipGetLastError());
}

// The column-major side of these three is in the CALLER's row numbering, the interleaved side in the library's (tiles.h):
// perm[i] = caller's row of internal row i (null: same numbering).
// out = a x + b y (or a conj(x) + b y) for one vector (out may alias x or y): the column updates of the device-resident multivectors (wae_slot_axpby)
__global__ __launch_bounds__(256) void axpby1_kernel(cplx a, const cplx *x, cplx b, const cplx *y, cplx *out, size_t n, int conj_x) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        cplx r = cmul(a, conj_x ? cconj(x[e]) : x[e]);
        cfma(r, b, y[e]);
        out[e] = r;
    }
}
void launch_axpby1(cplx a, const cplx *x, cplx b, const cplx *y, cplx *out, size_t n, hipStream_t st, int conj_x) {
    if (!n) return;
    hipLaunchKernelGGL(axpby1_kernel, dim3(grid_for(n)), dim3(256), 0, st, a, x, b, y, out, n, conj_x);
    HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void colmajor_to_inter_kernel(const cplx *__restrict__ Xc, int64_t d, int r, cplx *__restrict__ Xi, int nb,
                                                                const int *__restrict__ perm) {
    const size_t total = (size_t)d * nb;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / nb;
        const int b = (int)(e - row * nb);
        const size_t src = perm ? (size_t)perm[row] : row;
        Xi[e] = (b < r) ? Xc[(size_t)b * d + src] : cplx{0.0, 0.0};
    }
}
void launch_colmajor_to_inter(const cplx *Xc, int64_t d, int r, cplx *Xi, int nb, hipStream_t st, const int *perm) {
    hipLaunchKernelGGL(colmajor_to_inter_kernel, dim3(grid_for((size_t)d * nb)), dim3(256), 0, st, Xc, d, r, Xi, nb, perm);
    HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void inter_to_colmajor_kernel(const cplx *__restrict__ Xi, int nb, int64_t d, int r, cplx *__restrict__ Xc,
                                                                const int *__restrict__ perm) {
    const size_t total = (size_t)d * r;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t b = e / d, row = e - b * d;
        const size_t dst = perm ? (size_t)perm[row] : row;
        Xc[b * d + dst] = Xi[row * nb + b];
    }
}
void launch_inter_to_colmajor(const cplx *Xi, int nb, int64_t d, int r, cplx *Xc, hipStream_t st, const int *perm) {
    if (!d || !r) return;
    hipLaunchKernelGGL(inter_to_colmajor_kernel, dim3(grid_for((size_t)d * r)), dim3(256), 0, st, Xi, nb, d, r, Xc, perm);
    HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(256) void replicate_kernel(const cplx *__restrict__ Vc, int64_t d, int l, cplx *__restrict__ Xi, int nb,
                                                        const int *__restrict__ perm) {
    const size_t total = (size_t)d * nb;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / nb;
        const int b = (int)(e - row * nb);
        const size_t src = perm ? (size_t)perm[row] : row;
        Xi[e] = Vc[(size_t)(b % l) * d + src];
    }
}
void launch_replicate(const cplx *Vc, int64_t d, int l, cplx *Xi, int nb, hipStream_t st, const int *perm) {
    hipLaunchKernelGGL(replicate_kernel, dim3(grid_for((size_t)d * nb)), dim3(256), 0, st, Vc, d, l, Xi, nb, perm);
    HIP_CHECK(hipGetLastError());
}

// A[(p*lA + c0 + c)*d + row] += sum_s w[s] z[s]^p X[row][s*l+c]
// X is interleaved [row][nb] and A column-major [row fastest]: a workgroup stages TR rows of X in LDS (coalesced 16-B reads),
// then every thread owns one (row, c) of the tile, sums over the systems in registers and touches each moment entry once
// (coalesced along the rows).  The first version read X with a 1-KB lane stride and re-read/re-wrote A once per system:
// 1.5 ms per chunk at 1M DoF against 0.4 ms of traffic.
constexpr int ACC_MAXP = 8;         // moments (2K) accumulated in registers per pass over the systems
__global__ __launch_bounds__(256) void beyn_accum_kernel(const cplx *__restrict__ Xi, int nb, int64_t d, int l, int nsys,
                                                         const cplx *__restrict__ w, const cplx *__restrict__ z, int npow, cplx *__restrict__ A,
                                                         int lA, int c0, int TR, const int *__restrict__ perm) {
    extern __shared__ cplx tile[];                           // TR x (nb + 1): the pad keeps the column reads off one bank
    const int ld = nb + 1;
    const int tid = threadIdx.x;
    for (int64_t row0 = (int64_t)blockIdx.x * TR; row0 < d; row0 += (int64_t)gridDim.x * TR) {
        for (int e = tid; e < TR * nb; e += 256) {
            const int r = e / nb, col = e - r * nb;
            tile[r * ld + col] = (row0 + r < d) ? Xi[(size_t)(row0 + r) * nb + col] : cplx{0.0, 0.0};
        }
        __syncthreads();
        for (int idx = tid; idx < TR * l; idx += 256) {
            const int c = idx / TR, r = idx - c * TR;
            const int64_t row = row0 + r;
            if (row >= d) continue;
            for (int p0 = 0; p0 < npow; p0 += ACC_MAXP) {
                const int np = npow - p0 < ACC_MAXP ? npow - p0 : ACC_MAXP;
                cplx acc[ACC_MAXP];
#pragma unroll
                for (int p = 0; p < ACC_MAXP; ++p) acc[p] = cplx{0.0, 0.0};
                for (int s = 0; s < nsys; ++s) {
                    cplx t = cmul(w[s], tile[r * ld + s * l + c]);
                    const cplx zs = z[s];
                    for (int q = 0; q < p0; ++q) t = cmul(t, zs);
#pragma unroll
                    for (int p = 0; p < ACC_MAXP; ++p) {
                        if (p < np) { acc[p].x += t.x; acc[p].y += t.y; t = cmul(t, zs); }
                    }
                }
#pragma unroll
                for (int p = 0; p < ACC_MAXP; ++p) {
                    if (p < np) {
                        cplx *dst = A + ((size_t)(p0 + p) * lA + c0 + c) * d + (perm ? (int64_t)perm[row] : row);   // the moments are the caller's
                        const cplx a = *dst;
                        *dst = cplx{a.x + acc[p].x, a.y + acc[p].y};
                    }
                }
            }
        }
        __syncthreads();
    }
}
void launch_beyn_accum(const cplx *Xi, int nb, int64_t d, int l, int nsys, const cplx *w, const cplx *z, int npow, cplx *A, hipStream_t st,
                       int lA, int c0, const int *perm) {
    if (lA <= 0) lA = l;
    if (!d || nb < 1) return;
    if (nb > 256) throw WaeError(WAE_ERR_INVALID, "beyn_accum: nb must be in 1..256");
    int TR = 2048 / (nb + 1);                                // <= 32 KB of LDS
    if (TR > 32) TR = 32;
    if (TR < 1) TR = 1;
    const int64_t tiles = (d + TR - 1) / TR;
    const unsigned grid = (unsigned)std::min<int64_t>(tiles, 4096);
    hipLaunchKernelGGL(beyn_accum_kernel, dim3(grid), dim3(256), (size_t)TR * (nb + 1) * sizeof(cplx), st, Xi, nb, d, l, nsys, w, z, npow, A, lA, c0, TR, perm);
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
// The small-matrix half of the lock-step GMRES on the device (lib.hip gmres_wide): one thread per column keeps that column's
// Hessenberg column, Givens rotations, residual estimate and convergence flags in HBM, so that no iteration ends in a
// device-to-host copy + stream synchronisation (round 1: 51 us of host turnaround per lock-step iteration).  The arithmetic is
// the host loop's, statement for statement (same rotations, same tests), hence the same iterates.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gmres_init_kernel(GmresDev S, const cplx *__restrict__ beta, const unsigned char *__restrict__ done, int use_mask) {
    __shared__ int act[256];
    const int b = threadIdx.x;
    int a = 0;
    if (b < S.nb) {
        S.g[b] = cplx{beta[b].x, 0.0};
        S.sv[b] = 1.0;
        S.vsq[b] = cplx{1.0, 0.0};
        S.conv[b] = done[b] ? 1 : 0;
        S.steps[b] = 0;
        a = done[b] ? 0 : 1;
    }
    act[b] = a;
    __syncthreads();
    if (b < (S.nb + 7) / 8) {
        int any = 0;
        for (int k = 0; k < 8; ++k) any |= act[b * 8 + k];
        S.cmask[b] = use_mask ? (unsigned char)any : (unsigned char)1;
    }
    if (b == 0) {
        int n = 0;
        for (int k = 0; k < S.nb; ++k) n += act[k];
        S.status[0] = n; S.status[1] = 0; S.status[2] = 0;
    }
}

// after Arnoldi step j: hd[i][b], i <= j: s_i^2-scaled dots of the new vector against the unnormalised basis; hd[j+1][b].x: norm of
// the orthogonalised vector (lib.hip: "lazy" basis).  Produces the Hessenberg column of the normalised recurrence, rotates it,
// updates g, the residual estimate and the flags; marks vectors whose running scale left [1/lim, lim] for renormalisation.
__global__ __launch_bounds__(256) void gmres_step_kernel(GmresDev S, const cplx *__restrict__ hd, int j, double tol, double lim, int use_mask,
                                                         const cplx *__restrict__ rn) {
    __shared__ int act[256];
    const int b = threadIdx.x;
    const int nb = S.nb, m = S.m;
    int a = 0;
    if (b < nb) {
        const int nvj = j + 1;
        const double sj = S.sv[(size_t)j * nb + b];
        const double r = rn ? rn[b].x : hd[(size_t)nvj * nb + b].x;        // norm of the new (unnormalised) vector
        if (S.Hraw) {                                         // the unnormalised recurrence itself (read by the pair steps):
            cplx *Hr = S.Hraw + (size_t)j * (m + 1) * nb;     // Op v_j = sum_{i<=j} Hraw[j][i] v_i + sub[j] v_{j+1}
            for (int i = 0; i <= j; ++i) Hr[(size_t)i * nb + b] = hd[(size_t)i * nb + b];
        }
        double svn = r > 0.0 ? 1.0 / r : 0.0;
        const bool resc = svn > lim || (svn > 0.0 && svn < 1.0 / lim);
        S.rescale[b] = resc ? cplx{r, 0.0} : cplx{0.0, 0.0};
        if (S.sub) S.sub[(size_t)j * nb + b] = resc ? r : 1.0;
        if (resc) {                                          // the vector is normalised in place by gmres_rescale_kernel
            S.vsq[(size_t)nvj * nb + b] = cplx{svn > 0.0 ? 1.0 : 0.0, 0.0};
            svn = svn > 0.0 ? 1.0 : 0.0;
            atomicOr(&S.status[2], 1);
        }
        S.sv[(size_t)nvj * nb + b] = svn;
        if (!S.conv[b]) {
            cplx *Hc = S.R + (size_t)j * (m + 1) * nb;       // column j: entries i = 0..j+1 at Hc[i*nb + b]
            for (int i = 0; i <= j; ++i) {
                const double si = S.sv[(size_t)i * nb + b];
                const double f = si > 0.0 ? sj / si : 0.0;
                const cplx c = hd[(size_t)i * nb + b];
                Hc[(size_t)i * nb + b] = cplx{c.x * f, c.y * f};
            }
            Hc[(size_t)(j + 1) * nb + b] = cplx{sj * r, 0.0};
            for (int i = 0; i < j; ++i) {
                const cplx aa = Hc[(size_t)i * nb + b], bb = Hc[(size_t)(i + 1) * nb + b];
                const double c = S.cs[(size_t)i * nb + b];
                const cplx sn = S.sn[(size_t)i * nb + b];
                const cplx sb = cmul(sn, bb), ca = cmul(cconj(sn), aa);
                Hc[(size_t)i * nb + b] = cplx{c * aa.x + sb.x, c * aa.y + sb.y};
                Hc[(size_t)(i + 1) * nb + b] = cplx{-ca.x + c * bb.x, -ca.y + c * bb.y};
            }
            const cplx av = Hc[(size_t)j * nb + b];
            const double bv = Hc[(size_t)(j + 1) * nb + b].x;
            const double aabs = hypot(av.x, av.y);
            const double t = sqrt(aabs * aabs + bv * bv);
            if (!(t > 0.0) || isnan(t)) {
                S.conv[b] = 1;
                if (isnan(t)) atomicOr(&S.status[1], 1);
            } else {
                double c;
                cplx sn;
                if (aabs == 0.0) { c = 0.0; sn = cplx{1.0, 0.0}; }
                else { c = aabs / t; const double q = bv / t; sn = cplx{av.x / aabs * q, av.y / aabs * q}; }
                S.cs[(size_t)j * nb + b] = c;
                S.sn[(size_t)j * nb + b] = sn;
                Hc[(size_t)j * nb + b] = cplx{c * av.x + sn.x * bv, c * av.y + sn.y * bv};
                Hc[(size_t)(j + 1) * nb + b] = cplx{0.0, 0.0};
                const cplx gj = S.g[(size_t)j * nb + b];
                const cplx gn = cmul(cconj(sn), gj);
                S.g[(size_t)(j + 1) * nb + b] = cplx{-gn.x, -gn.y};
                S.g[(size_t)j * nb + b] = cplx{c * gj.x, c * gj.y};
                S.steps[b] = j + 1;
                S.iters[b] += 1;
                const double rr = hypot(gn.x, gn.y) / S.bnorm[b];
                S.relres[b] = rr;
                if (isnan(rr)) atomicOr(&S.status[1], 1);
                const int hs = S.histlen[b];
                if (hs < S.histcap) { S.hist[(size_t)hs * nb + b] = rr; S.histlen[b] = hs + 1; }
                const int hn = hs + 1;
                if (rr <= 0.7 * tol) S.conv[b] = 1;
                else if (hn > 60 && hs < S.histcap && rr > 0.9 * S.hist[(size_t)(hn - 31) * nb + b]) { S.conv[b] = 1; S.stalled[b] = 1; }   // attainable accuracy reached
                else a = 1;
            }
        }
    }
    act[b] = a;
    __syncthreads();
    if (b < (nb + 7) / 8 && use_mask) {
        int any = 0;
        for (int k = 0; k < 8; ++k) any |= act[b * 8 + k];
        S.cmask[b] = (unsigned char)any;
    }
    if (b == 0) {
        int n = 0;
        for (int k = 0; k < nb; ++k) n += act[k];
        S.status[0] = n;
    }
}

// y = R^-1 g per column over that column's steps; out[i][b] = s_i y_i (coefficients against the unnormalised basis), 0 beyond
__global__ __launch_bounds__(256) void gmres_solve_y_kernel(GmresDev S, int ju, cplx *__restrict__ out) {
    const int b = threadIdx.x;
    const int nb = S.nb, m = S.m;
    if (b >= nb) return;
    const int k = S.steps[b];
    for (int i = k; i < ju; ++i) out[(size_t)i * nb + b] = cplx{0.0, 0.0};
    for (int i = k - 1; i >= 0; --i) {
        cplx sacc = S.g[(size_t)i * nb + b];
        for (int q = i + 1; q < k; ++q) {
            const cplx hq = S.R[((size_t)q * (m + 1) + i) * nb + b];
            const cplx yq = out[(size_t)q * nb + b];
            sacc.x -= hq.x * yq.x - hq.y * yq.y;
            sacc.y -= hq.x * yq.y + hq.y * yq.x;
        }
        const cplx dg = S.R[((size_t)i * (m + 1) + i) * nb + b];
        out[(size_t)i * nb + b] = (dg.x != 0.0 || dg.y != 0.0) ? cdiv(sacc, dg) : cplx{0.0, 0.0};
    }
    for (int i = 0; i < k; ++i) {
        const double f = S.sv[(size_t)i * nb + b];
        cplx y = out[(size_t)i * nb + b];
        out[(size_t)i * nb + b] = cplx{f * y.x, f * y.y};
    }
}

// columns flagged by gmres_step_kernel: V[row][b] /= factor[b]  (every workgroup leaves at once when no column is flagged)
__global__ __launch_bounds__(256) void gmres_rescale_kernel(cplx *__restrict__ V, const cplx *__restrict__ factor, const int *__restrict__ status, size_t total, int nb) {
    if (!status[2]) return;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const double f = factor[e % nb].x;
        if (f > 0.0) { const cplx x = V[e]; V[e] = cplx{x.x / f, x.y / f}; }
    }
}
__global__ void gmres_clear_rescale_kernel(int *status) { status[2] = 0; }

void launch_gmres_init(const GmresDev &S, const cplx *beta, const unsigned char *done, int use_mask, hipStream_t st) {
    hipLaunchKernelGGL(gmres_init_kernel, dim3(1), dim3(256), 0, st, S, beta, done, use_mask);
    HIP_CHECK(hipGetLastError());
}
// One thread per column, ahead of the update pass of a pair step (after dots2): alpha = u1^H u2 / u1^H u1 from the Gram entries and
// the coefficients (u_k = w_k - V c_k), c2m = c2 - alpha c1, and the coefficients hd2 of Op v_{j+1} against the unnormalised basis
// v_0..v_{j+1}:  Op v_{j+1} = Op (w1 - V c1) = w2 - sum_k c1_k Op v_k,  w2 = V c2 + alpha v_{j+1} + v_{j+2}  and
// Op v_k = sum_{i<=k} Hraw[k][i] v_i + sub[k] v_{k+1} (column j of it being c1 itself, with sub[j] = 1).
__global__ __launch_bounds__(256) void gmres_pair_coef_kernel(GmresDev S, int j, const cplx *__restrict__ c1, const cplx *__restrict__ c2,
                                                              const cplx *__restrict__ gram, cplx *__restrict__ alpha, cplx *__restrict__ c2m,
                                                              cplx *__restrict__ hd2) {
    const int b = threadIdx.x;
    const int nb = S.nb, m = S.m;
    if (b >= nb) return;
    double uu = gram[b].x;
    cplx u12 = gram[(size_t)nb + b];
    for (int i = 0; i <= j; ++i) {
        const double q = S.vsq[(size_t)i * nb + b].x;
        const double w = q > 0.0 ? 1.0 / q : 0.0;             // ||v_i||^2
        const cplx a = c1[(size_t)i * nb + b], c = c2[(size_t)i * nb + b];
        uu -= (a.x * a.x + a.y * a.y) * w;
        u12.x -= (a.x * c.x + a.y * c.y) * w;                 // conj(a) c
        u12.y -= (a.x * c.y - a.y * c.x) * w;
    }
    cplx al = {0.0, 0.0};
    if (uu > 0.0 && uu > 1e-28 * gram[b].x) al = cplx{u12.x / uu, u12.y / uu};
    alpha[b] = al;
    cplx *Hj = S.Hraw + (size_t)j * (m + 1) * nb;
    for (int i = 0; i <= j; ++i) {
        const cplx a = c1[(size_t)i * nb + b], c = c2[(size_t)i * nb + b];
        Hj[(size_t)i * nb + b] = a;
        c2m[(size_t)i * nb + b] = cplx{c.x - (al.x * a.x - al.y * a.y), c.y - (al.x * a.y + al.y * a.x)};
    }
    S.sub[(size_t)j * nb + b] = 1.0;
    for (int i = 0; i <= j + 1; ++i) {
        cplx t = i <= j ? c2[(size_t)i * nb + b] : al;
        for (int k = i; k <= j; ++k) {
            const cplx hk = S.Hraw[((size_t)k * (m + 1) + i) * nb + b], ck = c1[(size_t)k * nb + b];
            t.x -= hk.x * ck.x - hk.y * ck.y;
            t.y -= hk.x * ck.y + hk.y * ck.x;
        }
        if (i >= 1) {
            const double sb = S.sub[(size_t)(i - 1) * nb + b];
            const cplx ck = c1[(size_t)(i - 1) * nb + b];
            t.x -= sb * ck.x;
            t.y -= sb * ck.y;
        }
        hd2[(size_t)i * nb + b] = t;
    }
}
void launch_gmres_pair_coef(const GmresDev &S, int j, const cplx *c1, const cplx *c2, const cplx *gram, cplx *alpha, cplx *c2m, cplx *hd2, hipStream_t st) {
    hipLaunchKernelGGL(gmres_pair_coef_kernel, dim3(1), dim3(256), 0, st, S, j, c1, c2, gram, alpha, c2m, hd2);
    HIP_CHECK(hipGetLastError());
}
void launch_gmres_step(const GmresDev &S, const cplx *hd, int j, double tol, double lim, int use_mask, cplx *Vnew, int64_t n, hipStream_t st,
                       const cplx *rn) {
    hipLaunchKernelGGL(gmres_step_kernel, dim3(1), dim3(256), 0, st, S, hd, j, tol, lim, use_mask, rn);
    hipLaunchKernelGGL(gmres_rescale_kernel, dim3(512), dim3(256), 0, st, Vnew, S.rescale, S.status, (size_t)n * S.nb, S.nb);
    hipLaunchKernelGGL(gmres_clear_rescale_kernel, dim3(1), dim3(1), 0, st, S.status);
    HIP_CHECK(hipGetLastError());
}
void launch_gmres_solve_y(const GmresDev &S, int ju, cplx *out, hipStream_t st) {
    hipLaunchKernelGGL(gmres_solve_y_kernel, dim3(1), dim3(256), 0, st, S, ju, out);
    HIP_CHECK(hipGetLastError());
}

// X[row][t] = sum_i G[i][t] * V_i[row]   (tall-skinny product of the perturbation regrouping; V_i = V + i*stride)
__global__ __launch_bounds__(256) void gemv_multi_kernel(const cplx *__restrict__ V, size_t stride, int k, const cplx *__restrict__ G,
                                                         cplx *__restrict__ X, int64_t d, int T) {
    const size_t total = (size_t)d * T;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / T;
        const int t = (int)(e - row * T);
        cplx acc = {0.0, 0.0};
        for (int i = 0; i < k; ++i) cfma(acc, G[(size_t)i * T + t], V[(size_t)i * stride + row]);
        X[e] = acc;
    }
}
void launch_gemv_multi(const cplx *V, size_t stride, int k, const cplx *G, cplx *X, int64_t d, int T, hipStream_t st) {
    hipLaunchKernelGGL(gemv_multi_kernel, dim3(grid_for((size_t)d * T)), dim3(256), 0, st, V, stride, k, G, X, d, T);
    HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void triad_kernel(double2 *__restrict__ a, const double2 *__restrict__ b, const double2 *__restrict__ c, double s, size_t n2) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n2; e += (size_t)gridDim.x * 256) {
        double2 x = b[e], y = c[e];
        a[e] = double2{x.x + s * y.x, x.y + s * y.y};
    }
}
void launch_triad(double *a, const double *b, const double *c, double s_, int64_t n, hipStream_t st, unsigned grid_cap) {
    size_t n2 = (size_t)n / 2;
    hipLaunchKernelGGL(triad_kernel, dim3(grid_for(n2, grid_cap)), dim3(256), 0, st, (double2 *)a, (const double2 *)b, (const double2 *)c, s_, n2);
    HIP_CHECK(hipGetLastError());
}
