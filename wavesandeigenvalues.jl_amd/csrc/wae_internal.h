// Internal declarations shared by the translation units of libwaehip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/waehip.h"

typedef std::complex<double> zc;
typedef double2 cplx;   // device complex: x = re, y = im

struct WaeError : std::runtime_error {
    int code;
    WaeError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
void wae_set_error(const std::string &m);
int wae_internal_device(const wae_family *h);          // (mgpu.hip drives several handles through the C ABI)
hipStream_t wae_internal_stream(const wae_family *h);

#define HIP_CHECK(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            throw WaeError(WAE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " at " +   \
                                            __FILE__ + ":" + std::to_string(__LINE__));                  \
    } while (0)
#define WAE_REQUIRE(cond, msg)                                  \
    do {                                                        \
        if (!(cond)) throw WaeError(WAE_ERR_INVALID, (msg));    \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// host-side sparse matrices
// ---------------------------------------------------------------------------------------------------
struct CsrZ {                       // complex CSR, 0-based, sorted columns, no duplicates
    int64_t n = 0, m = 0;           // rows, cols
    std::vector<int> ptr, col;
    std::vector<zc> val;
    int64_t nnz() const { return (int64_t)col.size(); }
};
struct CsrD {                       // real CSR
    int64_t n = 0, m = 0;
    std::vector<int> ptr, col;
    std::vector<double> val;
    int64_t nnz() const { return (int64_t)col.size(); }
};
CsrZ csr_transpose(const CsrZ &A);
CsrD csr_transpose(const CsrD &A);
bool csr_same_pattern(const CsrZ &A, const CsrZ &B);
CsrZ galerkin(const CsrD &R, const CsrZ &A, const CsrD &P);   // R*A*P

// ---------------------------------------------------------------------------------------------------
// device-side operator description
// ---------------------------------------------------------------------------------------------------
constexpr int WAE_MAXG = 24;        // pattern groups per level operator
constexpr double WAE_LEVEL_SYM_TOL = 1e-14;   // symmetry tolerance of the hierarchy's OWN operators (Galerkin products: part of the preconditioner only)
constexpr int WAE_MAXP = 64;       // value planes in total per level operator

struct GroupDev {                   // one sparsity pattern shared by `nplanes` value planes
    const int *rowptr;              // n+1
    const int *col;                 // nnz
    const void *vals;               // [nnz][nplanes] interleaved, double (is_real) or double2
    int nplanes;
    int is_real;
    int plane0;                     // first plane index in the per-system coefficient table
    int conj_vals;                  // conjugate complex values on the fly (op = C on a symmetric pattern)
};
// tile-local storage of an operator (tiles.h): rows renumbered so that <= 256 consecutive rows form a compact brick of the
// mesh graph whose distinct columns (the "window") fit LDS
struct TileGroupDev {
    const int *sptr;                // 8*ntiles+1: entry offset of the slice of (tile, wavefront): 64/lpr rows, lpr lanes per row
    const unsigned short *sidx;     // window-local column of every entry ([k][lane] inside a slice)
    const void *svals;              // [entry][nplanes] double or double2, same value layout as GroupDev::vals
    const unsigned short *dslot;    // [rows]: window slot of the row's own column (its diagonal entry), 0xFFFF if it has none; may be null
};
struct TileDev {
    int ntiles;
    int wmax;                       // largest window
    int lpr;                        // lanes per row: 2 (tiles of up to 256 rows) or 4 (up to 128 rows; 256 with 16 wavefronts)
    int nwaves;                     // wavefronts per workgroup (slices per tile): 8, or 16 (lpr = 4 on the fine level)
    int nbuf;                       // window buffers in LDS: 2 (windows up to 608 rows) or 3 (up to 400: two windows in flight; lpr = 2 only)
    int unit;                       // the operator is plane 0 itself (coefficients 1, 0: the restriction), no coefficient table is read
    const int *row_ptr;             // ntiles+1
    const int *win_ptr;             // ntiles+1
    const int *win_cols;            // global (new) column of every window slot, ascending per tile
    unsigned *counters;             // 16 words, zero between launches: [k] next position of share k, [8] workgroups that have left
    TileGroupDev g0;                // the bulk group (two real planes on one pattern)
    // Every other group (boundary, flame, ... : a few entries in a few per cent of the rows) stays out of the tile kernel: a
    // pre-kernel sums their rows ("side rows") into side_acc, the tile kernel adds that in.
    int nside;
    const int *side_of_row;         // [n]: position in the side-row list, or -1
    const int *side_ptr, *side_col, *side_slot;      // CSR over the side rows; slot = plane index in the coefficient table
    const cplx *side_val;
    cplx *side_acc;                 // [nside][nb]
    // side rows with more than WAE_LONG_ROW entries (the transposed orientation of a flame term: the reference nodes' rows hold one
    // entry per flame node) are summed by a workgroup each (spmv_side_long_kernel) into their side_acc row; their CSR rows are empty
    int nlong_side;
    const int *ls_ptr, *ls_col, *ls_slot, *ls_side;   // per long side row: entry offsets; per entry: column, plane slot; side-row index
    const cplx *ls_val;
    cplx *ls_part;                  // [nlong_side][WAE_LONG_SPLIT][nb max] partial sums of the pieces of the long side rows
};
struct OpDev {
    int ngroups;
    int nplanes_total;
    int64_t n;
    GroupDev g[WAE_MAXG];
    const cplx *diag;               // [n][nplanes_total] diagonal of every plane (for Jacobi), may be null
    int conj_diag;                  // op = C: use conj(diag) (coefficients arrive already conjugated)
    const TileDev *tiles;           // HOST pointer (read by launch_spmv only): tile-local storage valid for this orientation, or null
    // Long rows (more than WAE_LONG_ROW entries in a group: e.g. the reference nodes' rows of the transposed flame term, one
    // entry per flame node) are taken out of the groups' CSR arrays: a team of 8 lanes walking 5 000 dependent gathers held
    // the whole launch (100 ms per Krylov iteration of an adjoint solve at 1M DoF).  A pre-kernel sums them with a whole
    // workgroup per row into long_acc; the row kernels add that in.
    int nlong;
    const int *long_rows;           // sorted
    const int *long_ptr;            // nlong+1 offsets into the entries
    const int *long_col, *long_slot;   // per entry: column, plane slot (coefficient index)
    const cplx *long_val;           // per entry: value
    cplx *long_acc;                 // [nlong][nb max] scratch filled by launch_spmv's pre-kernels
    cplx *long_part;                // [nlong][WAE_LONG_SPLIT][nb max] partial sums of the pieces the rows are summed in
    int long_conj;                  // conjugate the entry values (op = C on complex planes)
};
constexpr int WAE_LONG_ROW = 256;
constexpr int WAE_LONG_SPLIT = 32;       // workgroups a long row's entries are split over (kernels.hip long_row_piece)

// device buffer: owns its allocation (freed on destruction, so an exception that leaves a C-ABI entry through guarded()
// releases every function-local buffer); movable, not copyable
template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        release();
        n = count;
        if (count) HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
    }
    void upload(const T *h, size_t count, hipStream_t s) {
        if (count > n) alloc(count);
        if (count) HIP_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

struct GroupHost {                  // a pattern group held on device, N and T orientation
    int nplanes = 0;
    bool is_real = true;
    bool symmetric = false;         // A^T == A for every plane: T orientation aliases N
    int64_t nnz = 0;
    int plane0 = 0;
    DevBuf<int> rowptr, col, rowptr_t, col_t;
    DevBuf<double> vals, vals_t;    // raw storage (nnz*nplanes*(1 or 2) doubles)
};

struct TileStore {                  // device arrays behind a TileDev
    DevBuf<int> row_ptr, win_ptr, win_cols;
    DevBuf<unsigned> counters;
    DevBuf<int> sptr;
    DevBuf<unsigned short> sidx, dslot;
    DevBuf<double> svals;
    DevBuf<int> side_of_row, side_ptr, side_col, side_slot;
    DevBuf<cplx> side_val, side_acc;
    TileDev dev;
    bool ready = false;
    bool all_symmetric = false;     // every group symmetric: the N-orientation tiles serve op = T/C as well
    // transposed orientation (round 3): the bulk group is symmetric, so the tile storage itself serves op = T/C; only the side rows
    // (rows of the other groups' TRANSPOSES) are their own
    DevBuf<int> t_side_of_row, t_side_ptr, t_side_col, t_side_slot, t_ls_ptr, t_ls_col, t_ls_slot, t_ls_side;
    DevBuf<cplx> t_side_val, t_side_acc, t_ls_val, t_ls_part;
    TileDev dev_t;
    bool ready_t = false;
};

struct LongRows {                   // device arrays of the long rows of one orientation (see OpDev)
    int n = 0;
    DevBuf<int> rows, ptr, col, slot;
    DevBuf<cplx> val, acc, part;
};

struct LevelOp {                    // sum_q pc[q] * plane_q  at one multigrid level
    int64_t n = 0;
    int nplanes = 0;
    std::vector<GroupHost> groups;
    DevBuf<cplx> diag;              // [n][nplanes]
    TileStore tiles;
    LongRows long_n, long_t;        // long rows in N orientation / in T orientation
    OpDev dev(int op) const;        // op: WAE_OP_N / T / C
    // host copies of the planes (kept for Galerkin products and dense coarse assembly)
    std::vector<CsrZ> planes;
};

// Prolongation by FINE tile (round 4).  The coarse rows a fine tile of 256 rows talks to (its "slots", ~130 with a smoothed
// prolongator) fit LDS at any chunk width, so x += P e becomes a stream over the fine multivector: a tile's slots of e are staged in
// LDS (16 KB per 8-column chunk), its fine rows stream through, and nothing else is large -- where prolong_add_kernel gathers ~5
// coarse rows per fine row from L2.  Deterministic.
struct XferTilesDev {
    int ntiles = 0, maxslots = 0, maxent = 0;   // (maxent: most transfer entries of one tile)
    int64_t nslots = 0, nf = 0, nc = 0;
    const int *row_ptr = nullptr;           // fine tile t owns fine rows row_ptr[t] .. row_ptr[t+1]-1 (<= 256)
    const int *tptr = nullptr;              // its slots: tptr[t] .. tptr[t+1]-1 (index into clist)
    const int *clist = nullptr;             // coarse row of a slot
    const int *pptr = nullptr;              // per fine row: entries pptr[i] .. pptr[i+1]-1
    const unsigned short *ploc = nullptr;   //   slot of the entry, local to the row's tile
    const double *pval = nullptr;
};
struct XferTiles {
    DevBuf<int> row_ptr, tptr, clist, pptr;
    DevBuf<unsigned short> ploc;
    DevBuf<double> pval;
    XferTilesDev dev;
    bool ready = false;
};
void launch_prolong_tiles(const XferTilesDev &T, const cplx *Xc, cplx *X, int nb, hipStream_t s, const unsigned char *cmask = nullptr);

struct Transfer {                   // P (n_fine x n_coarse) and R = P^T as single-plane real operators
    int64_t nf = 0, nc = 0;
    DevBuf<int> p_ptr, p_col, r_ptr, r_col;
    DevBuf<double> p_val, r_val;
    XferTiles ft;                   // the prolongation by fine tile (level 0 of a hierarchy whose fine level is renumbered into tiles)
    TileStore r_tiles;              // R in tile-local storage (both levels renumbered into tiles): unit coefficient, no side rows
    OpDev devP() const;
    OpDev devR() const;
};

// kernel launch wrappers (kernels.hip) -----------------------------------------------------------------
enum { MODE_AX = 0, MODE_RES = 1, MODE_JAC = 2, MODE_ADD = 3, MODE_AX_DS = 4, MODE_RES_DS = 5, MODE_AX_J0 = 6 };
// Y = f(A X) over columns [0,nb) of interleaved multivectors (leading dimension nb).
//   pc: [nsys][nplanes_total] plane coefficients; column b uses row b / cps.
//   MODE_AX : Y = A X            MODE_RES: Y = B - A X
//   MODE_JAC: Y = X + w/diag (B - A X)   (diag from op.diag and pc)      MODE_ADD: Y = B + A X
//   MODE_AX_DS: Y = (A X)/diag          MODE_RES_DS: Y = (B - A X)/diag   (row-scaled operator / residual)
//   MODE_AX_J0: Y = A X and, as a second OUTPUT through the B pointer, B = w/diag (A X): the product and the first
//               (zero-guess) Jacobi sweep of the V-cycle applied to it, in one pass
// cmask (optional): one byte per 8-column chunk, 0 = chunk converged -> its columns are skipped (outputs untouched)
void launch_spmv(const OpDev &op, const cplx *pc, int cps, const cplx *X, cplx *Y, const cplx *B, double jac_w,
                 int nb, int mode, hipStream_t s, const unsigned char *cmask = nullptr);
// X = w/diag * B  (first Jacobi sweep from a zero initial guess)
void launch_jacobi0(const OpDev &op, const cplx *pc, int cps, const cplx *B, cplx *X, double jac_w, int nb, hipStream_t s,
                    const unsigned char *cmask = nullptr);
// multi-input variant: Y[:,0] = sum_q pc[q] plane_q X[:, term_of_plane(q)]  (X interleaved with leading dim nb)
void launch_spmv_multi(const OpDev &op, const cplx *pc, const int *plane_col, const cplx *X, cplx *Y, int nb, hipStream_t s);

// dense coarse level: assemble A_s = sum_q pc[s][q] planes[q] (n x n, row-major per system), invert in place
void launch_dense_assemble(const cplx *planes, int nplanes, int n, const cplx *pc, int nsys, int op,
                           cplx *Ainv, hipStream_t s);
void launch_dense_invert(cplx *Ainv, int n, int nsys, int *status, hipStream_t s);
void launch_dense_apply(const cplx *Ainv, int n, int cps, const cplx *X, cplx *Y, int nb, hipStream_t s, const unsigned char *cmask = nullptr);

// vector kernels on interleaved multivectors [n][nb]
void launch_fill_zero(cplx *X, size_t count, hipStream_t s);
void launch_copy(const cplx *X, cplx *Y, size_t count, hipStream_t s);
// partial dots: out[i][b] = sum_rows conj(V_i[row][b]) * W[row][b], i = 0..nv-1; V_i = V + i*stride
void launch_dots(const cplx *V, size_t stride, int nv, const cplx *W, int64_t n, int nb, cplx *partial, cplx *out, hipStream_t s,
                 const unsigned char *cmask = nullptr);
// W -= sum_i h[i][b] V_i
// W_j -= sum_i h[(i*cnt + j)*nb + b] V_i, j < cnt <= 4, one reading of the nv vectors (snapshot basis: block Gram-Schmidt update)
void launch_axpy_neg_multi(const cplx *V, size_t stride, int nv, const cplx *h, cplx *W, size_t wstride, int cnt, int64_t n, int nb, hipStream_t s);
void launch_axpy_neg(const cplx *V, size_t stride, int nv, const cplx *h, cplx *W, int64_t n, int nb, hipStream_t s,
                     const unsigned char *cmask = nullptr);
// Y = sum_i y[i][b] V_i
void launch_axpy_neg_norm(const cplx *V, size_t stride, int nv, const cplx *h, cplx *W, int64_t n, int nb, cplx *partial, cplx *norms,
                          hipStream_t s, const unsigned char *cmask = nullptr, const cplx *base = nullptr, cplx *inv_out = nullptr);
// launch_dots with every output (i, b) multiplied by scale[i][b].x
void launch_dots_scaled(const cplx *V, size_t stride, int nv, const cplx *W, int64_t n, int nb, cplx *partial, cplx *out, const cplx *scale,
                        hipStream_t s, const unsigned char *cmask = nullptr);
void launch_lincomb(const cplx *V, size_t stride, int nv, const cplx *y, cplx *Y, int64_t n, int nb, hipStream_t s);
void launch_lincomb_add(const cplx *V, size_t stride, int nv, const cplx *y, cplx *X, int64_t n, int nb, hipStream_t s);   // X += V y
// norms: out[b] = ||X[:,b]||_2  (real part of out[b], imag 0)
void launch_norms(const cplx *X, int64_t n, int nb, cplx *partial, cplx *out, hipStream_t s, const unsigned char *cmask = nullptr);
// Y[:,b] = X[:,b] * (1/alpha[b].x)  (0 if alpha tiny)
void launch_scale_inv(const cplx *X, const cplx *alpha, cplx *Y, int64_t n, int nb, hipStream_t s, const unsigned char *cmask = nullptr);
// out = a x + b y, one vector (out may alias x or y); conj_x: a conj(x) + b y
void launch_axpby1(cplx a, const cplx *x, cplx b, const cplx *y, cplx *out, size_t n, hipStream_t s, int conj_x = 0);
// Y += X
void launch_add(const cplx *X, cplx *Y, size_t count, hipStream_t s);
// layout changes: column-major d x r  <->  interleaved [d][nb] (columns >= r zero-filled / ignored)
// perm (optional, device): internal row i is the caller's row perm[i] (tiles.h) -- the column-major side is in the caller's numbering
void launch_colmajor_to_inter(const cplx *Xc, int64_t d, int r, cplx *Xi, int nb, hipStream_t s, const int *perm = nullptr);
void launch_inter_to_colmajor(const cplx *Xi, int nb, int64_t d, int r, cplx *Xc, hipStream_t s, const int *perm = nullptr);
// replicate V (col-major d x l) into an interleaved block: column b -> V[:, b % l]
void launch_replicate(const cplx *Vc, int64_t d, int l, cplx *Xi, int nb, hipStream_t s, const int *perm = nullptr);
// snapshot-basis helpers: one system's l columns out of a batch; X = sum_i y[i][b] Q_i[row][b % l]; zero selected columns
// V-cycle prolongation: X += P Xc (P real CSR, n fine rows)
void launch_prolong_add(const int *ptr, const int *col, const double *val, int64_t n, const cplx *Xc, cplx *X, int nb, hipStream_t s,
                        const unsigned char *cmask = nullptr);
// out[(i*nw + j)*nb + b] = V_i[:,b]^H W_j[:,b], nw <= 4: one pass over V for nw right-hand vectors
void launch_dots_multi(const cplx *V, size_t sv, int nv, const cplx *W, size_t sw, int nw, int64_t n, int nb, cplx *partial, cplx *out,
                       hipStream_t s);
void launch_gather_rows(const cplx *X, const int *rows, int64_t nrows, int nb, cplx *out, hipStream_t s);
void launch_scatter_add_rows(const cplx *D, const int *rows, int64_t nrows, int nb, cplx *X, hipStream_t s);
void launch_extract_cols(const cplx *X, int nb, int off, int l, cplx *out, int64_t n, hipStream_t s);
void launch_lincomb_rep(const cplx *Q, size_t stride, int nv, const cplx *y, cplx *X, int64_t n, int nb, int l, hipStream_t s);
void launch_mask_cols(cplx *X, const cplx *keep, int64_t n, int nb, hipStream_t s);
// Beyn accumulation: A[(p*lA+c0+c)*d + row] += sum_s w[s] z[s]^p X[row][s*l+c], s < nsys, p < npow  (lA columns in A; 0 = l)
void launch_beyn_accum(const cplx *Xi, int nb, int64_t d, int l, int nsys, const cplx *w, const cplx *z, int npow, cplx *A, hipStream_t s,
                       int lA = 0, int c0 = 0, const int *perm = nullptr);
// X[row][t] = sum_i G[i][t] V_i[row], V_i = V + i*stride (single vectors), X interleaved with leading dimension T
void launch_gemv_multi(const cplx *V, size_t stride, int k, const cplx *G, cplx *X, int64_t d, int T, hipStream_t s);
// device-resident state of the lock-step GMRES recurrence (one thread per column, kernels.hip gmres_*_kernel)
struct GmresDev {
    int nb, m, histcap;
    cplx *R;                // [m][m+1][nb]: rotated Hessenberg columns
    double *cs;             // [m][nb]
    cplx *sn;               // [m][nb]
    cplx *g;                // [m+1][nb]
    double *sv;             // [m+2][nb]: 1/norm of the unnormalised basis vectors (running products)
    cplx *vsq;              // [m+2][nb]: 1/||v_i||^2 (read by dots_scaled)
    int *conv, *steps, *iters, *histlen, *stalled;   // [nb]
    double *relres, *bnorm; // [nb]
    double *hist;           // [histcap][nb] residual history (stagnation test)
    cplx *rescale;          // [nb] factor of a pending renormalisation (0 = none)
    int *status;            // [0] active columns, [1] NaN seen, [2] renormalisation pending
    unsigned char *cmask;   // per 8-column chunk: any active column
    cplx *Hraw;             // [m][m+1][nb]: the unnormalised recurrence, Op v_k = sum_{i<=k} Hraw[k][i] v_i + sub[k] v_{k+1} (pair steps); may be null
    double *sub;            // [m][nb]
};
void launch_gmres_init(const GmresDev &S, const cplx *beta, const unsigned char *done, int use_mask, hipStream_t s);
void launch_gmres_step(const GmresDev &S, const cplx *hd, int j, double tol, double lim, int use_mask, cplx *Vnew, int64_t n, hipStream_t s,
                       const cplx *rn = nullptr);     // rn (optional): the norm of the new vector, instead of hd[j+1]
// pair steps (two Arnoldi steps per pass over the basis; kernels.hip)
void launch_dots2_scaled(const cplx *V, size_t stride, int nv, const cplx *W1, const cplx *W2, int64_t n, int nb, cplx *partial, cplx *out1,
                         cplx *out2, cplx *gram_out, const cplx *scale, hipStream_t s, const unsigned char *cmask = nullptr);
void launch_axpy2_norm(const cplx *V, size_t stride, int nv, const cplx *c1, const cplx *c2m, const cplx *alpha, cplx *W1, cplx *W2, int64_t n, int nb,
                       cplx *partial, cplx *norms, cplx *inv_out, hipStream_t s, const unsigned char *cmask = nullptr);
void launch_gmres_pair_coef(const GmresDev &S, int j, const cplx *c1, const cplx *c2, const cplx *gram, cplx *alpha, cplx *c2m, cplx *hd2, hipStream_t s);
void launch_gmres_solve_y(const GmresDev &S, int ju, cplx *out, hipStream_t s);
// triad for bandwidth measurement
void launch_triad(double *a, const double *b, const double *c, double s, int64_t n, hipStream_t st, unsigned grid_cap = 8192);
