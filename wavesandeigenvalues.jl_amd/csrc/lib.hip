// libwaehip.so -- family handle, multigrid-preconditioned batched GMRES, Beyn moment loop, C ABI.
// gfx950 only.  See include/waehip.h for the contract of every exported function.
#include <algorithm>
#include <atomic>
#include <exception>
#include <future>
#include <chrono>
#include <cmath>
#include <map>
#include <memory>
#include <tuple>

#include "amg.h"
#include "tiles.h"
#include "wae_internal.h"

static thread_local std::string g_last_error;
void wae_set_error(const std::string &m) { g_last_error = m; }

// ----------------------------------------------------------------------------------------------------
// level operators on device
// ----------------------------------------------------------------------------------------------------
OpDev LevelOp::dev(int op) const {
    OpDev o;
    memset(&o, 0, sizeof(o));
    o.ngroups = (int)groups.size();
    o.nplanes_total = nplanes;
    o.n = n;
    o.diag = diag.p;
    o.conj_diag = (op == WAE_OP_C) ? 1 : 0;
    o.tiles = !tiles.ready ? nullptr : (op == WAE_OP_N || tiles.all_symmetric) ? &tiles.dev : (tiles.ready_t ? &tiles.dev_t : nullptr);
    {
        const LongRows &LR = (op == WAE_OP_N) ? long_n : long_t;
        o.nlong = LR.n;
        o.long_rows = LR.rows.p; o.long_ptr = LR.ptr.p; o.long_col = LR.col.p; o.long_slot = LR.slot.p;
        o.long_val = LR.val.p; o.long_acc = LR.acc.p; o.long_part = LR.part.p;
        o.long_conj = (op == WAE_OP_C) ? 1 : 0;
    }
    for (size_t g = 0; g < groups.size(); ++g) {
        const GroupHost &G = groups[g];
        GroupDev &D = o.g[g];
        const bool tr = (op != WAE_OP_N) && !G.symmetric;
        D.rowptr = tr ? G.rowptr_t.p : G.rowptr.p;
        D.col = tr ? G.col_t.p : G.col.p;
        D.vals = tr ? (const void *)G.vals_t.p : (const void *)G.vals.p;
        D.nplanes = G.nplanes;
        D.is_real = G.is_real ? 1 : 0;
        D.plane0 = G.plane0;
        D.conj_vals = (op == WAE_OP_C && !G.is_real) ? 1 : 0;
    }
    return o;
}
static OpDev transfer_dev(const DevBuf<int> &ptr, const DevBuf<int> &col, const DevBuf<double> &val, int64_t n) {
    OpDev o;
    memset(&o, 0, sizeof(o));
    o.ngroups = 1;
    o.nplanes_total = 1;
    o.n = n;
    o.g[0].rowptr = ptr.p;
    o.g[0].col = col.p;
    o.g[0].vals = val.p;
    o.g[0].nplanes = 1;
    o.g[0].is_real = 1;
    o.tiles = nullptr;
    o.nlong = 0;
    return o;
}
OpDev Transfer::devP() const { return transfer_dev(p_ptr, p_col, p_val, nf); }
OpDev Transfer::devR() const {
    OpDev o = transfer_dev(r_ptr, r_col, r_val, nc);
    o.tiles = r_tiles.ready ? &r_tiles.dev : nullptr;
    return o;
}

struct RbState {                     // snapshot basis of wae_beyn_moments_rb (one per handle)
    cplx *Q = nullptr;               // store: cap snapshots of d x l (interleaved [row][column]); slots < S are orthonormal per column
    int cap = 0, l = 0, S = 0;
    std::vector<int> kact;           // terms that take part in the projection
    std::vector<zc> Hk;              // Hk[ki][(s*cap + i)*l + c] = q_i^H A_k q_s   (column c's basis)
    std::vector<zc> g;               // g[i*l + c] = q_i^H v_c
    DevBuf<cplx> W, Vi, hb, alpha, alpha2, ycoef;   // W_k = A_k Q (resident), probe columns interleaved, small scratch
    std::future<void> w_job;         // W (20 GB at 1M unknowns: ~0.4 s of hipMalloc) is mapped on a helper thread while the first
    void wait_w() { if (w_job.valid()) w_job.get(); }       // snapshot systems are solved; whoever touches W waits for it here
    bool vi_valid = false;           // Vi holds the probe matrix the basis was started with (false after an import)
    ~RbState() { if (w_job.valid()) w_job.wait(); }
};

struct wae_family {
    bool vc_light = false;               // the current solve belongs to the projected phase of a contour integral (1-5 steps from a good guess): vcycle() runs its light form
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t d = 0;
    int T = 0;
    std::vector<int> term_plane;     // term k -> plane index (in the order planes were discovered)
    std::vector<zc> term_scale;      // term k = scale * plane
    std::vector<int64_t> term_nnz;
    int nplanes = 0;
    std::vector<CsrZ> planes0;       // host copies of the fine planes (set-up input), in the library's row numbering
    // Row renumbering (tiles.h): internal row i is the caller's row perm[i].  Applied to the term matrices at create, to
    // every vector at the ABI boundary (layout kernels), never visible outside.
    std::vector<int> perm_h;
    DevBuf<int> perm_dev;
    std::vector<int> tile_row_ptr;   // tiles of the fine level (empty: no tiling)
    const int *perm() const { return perm_dev.p; }
    std::vector<LevelOp> ops;        // ops[0] = fine level
    std::vector<std::vector<int>> slot_plane;   // per level: slot -> plane
    std::vector<Transfer> xfer;
    // dense coarsest level
    int64_t nc = 0;
    DevBuf<cplx> dense_planes, Ainv;
    DevBuf<int> dstatus;
    bool solver_ready = false;
    double jac_w = 0.8;                  // weight of the pre-smoothing sweeps of the full V-cycle (opts[2])
    double jac_w_post = 0.9;             // ... of its post-smoothing sweeps (opts[10])
    double jac_w_light = 0.5;            // ... of the single sweep of the light cycle (opts[11]; projected phase of a contour integral)
    int nsweeps = 1, restart = 30, NB = 64;
    // workspaces
    std::vector<DevBuf<cplx>> lx, lb, lt;
    DevBuf<cplx> V, W, Z, Xs, Bs, U, partial, hdev, ydev, pcdev, one_dev, io_a, io_b, zw_dev;
    DevBuf<cplx> vsq;                // 1/||v_i||^2 per basis slot and column: the wide-batch GMRES keeps its basis unnormalised
    DevBuf<cplx> rbQ;                // library-owned snapshot store of wae_beyn_moments_rb
    RbState rb;                      // the snapshot basis and its projected terms
    DevBuf<int> plane_col_dev;
    DevBuf<unsigned char> cmask;     // one byte per 8-column chunk of the current batch (0 = converged)
    // device-resident recurrence of the wide-batch GMRES (gmres_wide)
    DevBuf<cplx> gs_R, gs_sn, gs_g, gs_rescale, gs_Hraw, gs_pair;
    DevBuf<double> gs_sub;
    DevBuf<double> gs_cs, gs_sv, gs_relres, gs_bnorm, gs_hist;
    DevBuf<int> gs_int;              // conv | steps | iters | histlen | stalled | status(4)
    DevBuf<unsigned char> gs_done;
    // penalty (Dirichlet-like) rows found at set-up: their sub-block as a small operator of its own (see penalty_polish)
    int64_t n_penalty = 0;
    LevelOp pen_op;
    std::vector<int> pen_slot;
    LevelOp pen_row_op;              // the penalty ROWS of the operator (n_penalty x d): their residual without a full SpMV
    std::vector<int> pen_row_slot;
    DevBuf<int> pen_rows;
    DevBuf<cplx> pen_b, pen_x, pen_t;
    // device-resident multivectors of the caller ("slots", wae_slot_*): d x ncols, column-major, in the library's row numbering
    struct Slot { DevBuf<cplx> buf; int ncols = 0; };
    Slot slots[WAE_NSLOTS];
    // work space of the Arnoldi processes (kept between calls); after wae_arnoldi_shiftinvert_slots the basis of that call stays in
    // arn_EV: arn_cols vectors of arn_nsys systems each, interleaved [row][system], for wae_arnoldi_ritz_to_slot
    DevBuf<cplx> arn_EV, arn_t, arn_pcM, arn_hcol, arn_stage, arn_gdir;
    int arn_nsys = 0, arn_cols = 0;
    DevBuf<cplx> pt_ws, pt_Gd, pt_pcd;   // work space of wae_perturb / wae_perturb_slots (kept between calls)
    cplx *h_pinned = nullptr;        // (restart+2)*NB
    cplx *h_pin_pair = nullptr;      // staging of the pair steps of the narrow batches (gmres)
    size_t h_pin_pair_n = 0;
    size_t pc_stride_level = 0;      // elements per level in pcdev
    ~wae_family() {                  // every DevBuf member frees itself
        if (stream) (void)hipStreamSynchronize(stream);
        if (h_pinned) (void)hipHostFree(h_pinned);
        if (h_pin_pair) (void)hipHostFree(h_pin_pair);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

int wae_internal_device(const wae_family *h) { return h->device; }
hipStream_t wae_internal_stream(const wae_family *h) { return h->stream; }

static bool plane_is_real(const CsrZ &A) {
    for (const zc &v : A.val)
        if (v.imag() != 0.0) return false;
    return true;
}

// Build the device representation of sum_q pc[q] plane_q from host planes; returns slot -> plane map.
// sym_tol: when is a plane "symmetric", i.e. applied in its stored orientation for op = T / C?  0: only if mirror entries are equal
// bit for bit (A' is exactly A').  > 0: if  |a_ij - a_ji| <= sym_tol * min(s_i, s_j),  s_i = the largest OFF-DIAGONAL magnitude of
// row i -- the rounding scale of a row's assembled sums that a penalty / Dirichlet diagonal entry cannot inflate.
static std::vector<int> build_levelop(LevelOp &L, const std::vector<CsrZ> &planes, hipStream_t st, double sym_tol) {
    L.n = planes.empty() ? 0 : planes[0].n;
    L.nplanes = (int)planes.size();
    struct Grp { std::vector<int> members; bool real; };
    std::vector<Grp> grps;
    for (int q = 0; q < (int)planes.size(); ++q) {
        const bool re = plane_is_real(planes[q]);
        bool placed = false;
        for (auto &g : grps)
            if (g.real == re && csr_same_pattern(planes[g.members[0]], planes[q])) { g.members.push_back(q); placed = true; break; }
        if (!placed) grps.push_back(Grp{{q}, re});
    }
    if ((int)grps.size() > WAE_MAXG) throw WaeError(WAE_ERR_INVALID, "too many distinct sparsity patterns (max 24)");
    if ((int)planes.size() > WAE_MAXP) throw WaeError(WAE_ERR_INVALID, "too many distinct term matrices (max 64)");
    std::vector<int> slot_plane;
    L.groups.clear();
    L.groups.resize(grps.size());
    struct LongEntries {
        std::map<int, std::vector<std::tuple<int, int, zc>>> rows;      // row -> (column, slot, value)
        void add(int r, int c, int slot, zc v) { rows[r].emplace_back(c, slot, v); }
    };
    LongEntries long_n, long_t;
    L.long_n = LongRows();
    L.long_t = LongRows();
    for (size_t gi = 0; gi < grps.size(); ++gi) {
        const Grp &g = grps[gi];
        GroupHost &G = L.groups[gi];
        const CsrZ &A0 = planes[g.members[0]];
        const int np = (int)g.members.size();
        G.nplanes = np;
        G.is_real = g.real;
        G.nnz = A0.nnz();
        G.plane0 = (int)slot_plane.size();
        for (int q : g.members) slot_plane.push_back(q);
        const int w = g.real ? 1 : 2;
        auto pack = [&](const std::vector<const CsrZ *> &mats, std::vector<double> &out) {
            const int64_t nnz = mats[0]->nnz();
            out.resize((size_t)nnz * np * w);
            const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(8, nnz / 262144));      // (host threads: entries in contiguous ranges)
            std::vector<std::future<void>> jobs;
            for (int t = 0; t < nth; ++t)
                jobs.push_back(std::async(nth > 1 ? std::launch::async : std::launch::deferred, [&, t]() {
                    const int64_t lo = nnz * t / nth, hi = nnz * (t + 1) / nth;
                    for (int64_t p = lo; p < hi; ++p)
                        for (int k = 0; k < np; ++k) {
                            const zc v = mats[k]->val[p];
                            if (g.real) out[(size_t)p * np + k] = v.real();
                            else { out[((size_t)p * np + k) * 2] = v.real(); out[((size_t)p * np + k) * 2 + 1] = v.imag(); }
                        }
                }));
            for (auto &j : jobs) j.get();
        };
        // long rows go to the level's long-row store (OpDev) and leave the group's CSR arrays
        auto strip = [&](const std::vector<const CsrZ *> &src, std::vector<CsrZ> &kept, LongEntries &LE) {
            kept.clear();
            std::vector<char> is_long(src[0]->n, 0);
            bool any = false;
            const int limit = getenv("WAE_LONG_ROW") ? std::max(1, atoi(getenv("WAE_LONG_ROW"))) : WAE_LONG_ROW;   // (tests lower it)
            for (int64_t i = 0; i < src[0]->n; ++i)
                if (src[0]->ptr[i + 1] - src[0]->ptr[i] > limit) { is_long[i] = 1; any = true; }
            if (!any) return false;
            for (size_t k = 0; k < src.size(); ++k) {
                const CsrZ &A = *src[k];
                CsrZ B;
                B.n = A.n; B.m = A.m;
                B.ptr.assign(A.n + 1, 0);
                for (int64_t i = 0; i < A.n; ++i) {
                    if (is_long[i]) {
                        for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) LE.add((int)i, A.col[p], G.plane0 + (int)k, A.val[p]);
                    } else {
                        B.col.insert(B.col.end(), A.col.begin() + A.ptr[i], A.col.begin() + A.ptr[i + 1]);
                        B.val.insert(B.val.end(), A.val.begin() + A.ptr[i], A.val.begin() + A.ptr[i + 1]);
                    }
                    B.ptr[i + 1] = (int)B.col.size();
                }
                kept.push_back(std::move(B));
            }
            return true;
        };
        std::vector<const CsrZ *> own;
        for (int q : g.members) own.push_back(&planes[q]);
        // transpose orientation.  Symmetry test: same pattern and mirror entries that agree exactly (sym_tol = 0) or to within sym_tol
        // of the smaller of the two rows' off-diagonal scales.  Why a tolerance exists at all: a finite-element matrix assembled in
        // floating point is symmetric only up to the order of its element sums (K and M of the 200k..1M-DoF annulus: mirror entries
        // differ by 1e-16 of the row scale in half of the positions), and the exact test sends every adjoint product of such a family
        // through a second, transposed copy of the operator and past the tile kernel.  A plane accepted with sym_tol > 0 is applied
        // in its stored orientation for op = T / C: the product then differs from the exact transposed one by that assembly
        // rounding.  The caller asks for it (wae_family_create_opts); the hierarchy's own coarse levels use 1e-14.
        // The mirror entry a_ji is looked up in row j (sorted columns) on the host threads; the transposed copies are built only for
        // a group that fails the test (or whose rows are not sorted: then the transpose decides, as it used to).
        auto row_scales = [](const CsrZ &P) {                    // largest off-diagonal magnitude per row
            std::vector<double> sc((size_t)P.n, 0.0);
            const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(16, P.n / 8192));
            std::vector<std::future<void>> jobs;
            for (int t = 0; t < nth; ++t)
                jobs.push_back(std::async(nth > 1 ? std::launch::async : std::launch::deferred, [&, t]() {
                    const int64_t lo = P.n * t / nth, hi = P.n * (t + 1) / nth;
                    for (int64_t i = lo; i < hi; ++i) {
                        double m = 0.0;
                        for (int p = P.ptr[i]; p < P.ptr[i + 1]; ++p)
                            if (P.col[p] != (int)i) m = std::max(m, std::norm(P.val[p]));
                        sc[(size_t)i] = std::sqrt(m);
                    }
                }));
            for (auto &j : jobs) j.get();
            return sc;
        };
        auto mirror_test = [sym_tol, &row_scales](const CsrZ &P) -> int {     // 1 symmetric, 0 not, -1 unsorted rows (undecided)
            if (P.n != P.m) return 0;
            std::vector<double> sc;
            if (sym_tol > 0.0) sc = row_scales(P);
            const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(16, P.n / 8192));
            std::vector<int> verdict(nth, 1);
            std::vector<std::future<void>> jobs;
            for (int t = 0; t < nth; ++t)
                jobs.push_back(std::async(nth > 1 ? std::launch::async : std::launch::deferred, [&, t]() {
                    const int64_t lo = P.n * t / nth, hi = P.n * (t + 1) / nth;
                    int vd = 1;                                // (thread-local: the per-thread slots share cache lines)
                    for (int64_t i = lo; i < hi && vd == 1; ++i)
                        for (int p = P.ptr[i]; p < P.ptr[i + 1]; ++p) {
                            if (p > P.ptr[i] && P.col[p - 1] >= P.col[p]) { vd = -1; break; }
                            const int j = P.col[p];
                            if (j == i) continue;
                            const int *b = P.col.data() + P.ptr[j], *e = P.col.data() + P.ptr[j + 1];
                            const int *f = std::lower_bound(b, e, (int)i);
                            if (f == e || *f != (int)i) { vd = 0; break; }
                            const zc m = P.val[(size_t)(f - P.col.data())];
                            if (m == P.val[p]) continue;
                            if (!(sym_tol > 0.0 && std::abs(P.val[p] - m) <= sym_tol * std::min(sc[(size_t)i], sc[(size_t)j]))) { vd = 0; break; }
                        }
                    verdict[t] = vd;
                }));
            for (auto &j : jobs) j.get();
            int v = 1;
            for (int t = 0; t < nth; ++t) { if (verdict[t] == -1) v = -1; else if (verdict[t] == 0 && v == 1) v = 0; }
            return v;
        };
        std::vector<CsrZ> tr;
        bool sym = (A0.n == A0.m);
        bool undecided = false;
        for (size_t k = 0; k < g.members.size() && sym; ++k) {
            const int v = mirror_test(planes[g.members[k]]);
            if (v == 0) sym = false;
            if (v < 0) { undecided = true; break; }
        }
        if (!sym || undecided) {
            std::vector<std::future<CsrZ>> tj;
            for (int q : g.members) tj.push_back(std::async(std::launch::async, [&planes, q]() { return csr_transpose(planes[q]); }));
            for (size_t k = 0; k < tj.size(); ++k) {
                const int q = g.members[k];
                tr.push_back(tj[k].get());
                if (!sym) continue;
                const CsrZ &P0 = planes[q], &P1 = tr.back();
                if (!(P1.ptr == P0.ptr && P1.col == P0.col)) { sym = false; continue; }
                std::vector<double> sc;
                if (sym_tol > 0.0) sc = row_scales(P0);
                for (int64_t i = 0; i < P0.n && sym; ++i)          // (same pattern: entry e of P1 is the mirror of entry e of P0)
                    for (int e = P0.ptr[i]; e < P0.ptr[i + 1]; ++e) {
                        if (P0.val[e] == P1.val[e]) continue;
                        const int j = P0.col[e];
                        if (!(sym_tol > 0.0 && j != i && std::abs(P0.val[e] - P1.val[e]) <= sym_tol * std::min(sc[(size_t)i], sc[(size_t)j]))) { sym = false; break; }
                    }
            }
        }
        G.symmetric = sym;
        std::vector<CsrZ> kept;
        const bool stripped_n = strip(own, kept, long_n);
        if (stripped_n && sym) {                             // the T orientation aliases these arrays: same rows, same entries
            std::vector<CsrZ> dummy;
            strip(own, dummy, long_t);
        }
        std::vector<const CsrZ *> mats;
        if (stripped_n) for (const CsrZ &M : kept) mats.push_back(&M);
        else mats = own;
        std::vector<double> packed;
        pack(mats, packed);
        G.rowptr.upload(mats[0]->ptr.data(), mats[0]->ptr.size(), st);
        G.col.upload(mats[0]->col.data(), mats[0]->col.size(), st);
        G.vals.upload(packed.data(), packed.size(), st);
        std::vector<CsrZ> kept_t;
        std::vector<double> tp;
        if (!sym) {
            std::vector<const CsrZ *> trp;
            for (const CsrZ &t : tr) trp.push_back(&t);
            const bool stripped_t = strip(trp, kept_t, long_t);
            std::vector<const CsrZ *> tm;
            if (stripped_t) for (const CsrZ &t : kept_t) tm.push_back(&t);
            else tm = trp;
            pack(tm, tp);
            G.rowptr_t.upload(tm[0]->ptr.data(), tm[0]->ptr.size(), st);
            G.col_t.upload(tm[0]->col.data(), tm[0]->col.size(), st);
            G.vals_t.upload(tp.data(), tp.size(), st);
        }
        HIP_CHECK(hipStreamSynchronize(st));   // host staging buffers die at scope end
    }
    auto upload_long = [&](const LongEntries &LE, LongRows &LR) {
        LR.n = (int)LE.rows.size();
        if (!LR.n) return;
        std::vector<int> rows, ptr(1, 0), col, slot;
        std::vector<cplx> val;
        for (const auto &kv : LE.rows) {
            rows.push_back(kv.first);
            for (const auto &e : kv.second) { col.push_back(std::get<0>(e)); slot.push_back(std::get<1>(e)); val.push_back(cplx{std::get<2>(e).real(), std::get<2>(e).imag()}); }
            ptr.push_back((int)col.size());
        }
        LR.rows.upload(rows.data(), rows.size(), st); LR.ptr.upload(ptr.data(), ptr.size(), st);
        LR.col.upload(col.data(), col.size(), st); LR.slot.upload(slot.data(), slot.size(), st);
        LR.val.upload(val.data(), val.size(), st);
        LR.acc.alloc((size_t)LR.n * 256);                     // batch widths up to 256 columns
        LR.part.alloc((size_t)LR.n * WAE_LONG_SPLIT * 256);
        HIP_CHECK(hipStreamSynchronize(st));
    };
    upload_long(long_n, L.long_n);
    upload_long(long_t, L.long_t);
    // diagonals [n][nplanes] in slot order
    std::vector<cplx> dg((size_t)L.n * L.nplanes, cplx{0.0, 0.0});
    for (int s = 0; s < L.nplanes; ++s) {
        const CsrZ &A = planes[slot_plane[s]];
        for (int64_t i = 0; i < A.n; ++i)
            for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p)
                if (A.col[p] == i) dg[(size_t)i * L.nplanes + s] = cplx{A.val[p].real(), A.val[p].imag()};
    }
    L.diag.upload(dg.data(), dg.size(), st);
    HIP_CHECK(hipStreamSynchronize(st));
    return slot_plane;
}

// tile-local storage of an operator whose rows have been cut into tiles (tiles.h): windows and the bulk group (two real planes on
// one pattern).  U: pattern that defines the windows (every column any plane of the operator touches).
static bool build_tiles_core(TileStore &T, const std::vector<const CsrZ *> &bulk, const Pattern &U, const std::vector<int> &row_ptr, int lpr,
                             hipStream_t st, const char *what, int nbuf = 2, int nwaves = 8) {
    T = TileStore();
    if (row_ptr.size() < 2) return false;
    const TileWindows W = build_windows(U, row_ptr);
    const int nt = (int)row_ptr.size() - 1;
    int wmax = 0;
    for (int t = 0; t < nt; ++t) wmax = std::max(wmax, W.win_ptr[t + 1] - W.win_ptr[t]);
    if (wmax > 65535) return false;
    T.row_ptr.upload(row_ptr.data(), row_ptr.size(), st);
    T.win_ptr.upload(W.win_ptr.data(), W.win_ptr.size(), st);
    T.win_cols.upload(W.win_cols.data(), W.win_cols.size(), st);
    memset(&T.dev, 0, sizeof(T.dev));
    {
        const TileGroupHost H = build_tile_group(bulk, true, row_ptr, W, lpr, nwaves);
        T.sptr.upload(H.sptr.data(), H.sptr.size(), st);
        T.sidx.upload(H.sidx.data(), H.sidx.size(), st);
        T.svals.upload(H.svals.data(), H.svals.size(), st);
        T.dslot.upload(H.dslot.data(), H.dslot.size(), st);
        HIP_CHECK(hipStreamSynchronize(st));                 // H dies at the end of this scope
        T.dev.g0.sptr = T.sptr.p;
        T.dev.g0.sidx = T.sidx.p;
        T.dev.g0.svals = T.svals.p;
        T.dev.g0.dslot = T.dslot.p;
        if (getenv("WAE_SETUP_DEBUG")) {
            int over = 0, full = 0;                          // slices longer than the register-resident entries per lane
            for (size_t i = 0; i + 1 < H.sptr.size(); ++i) over += (H.sptr[i + 1] - H.sptr[i]) / 64 > (nwaves == 16 ? 4 : (lpr == 2 ? 8 : 12));
            for (int t = 0; t < nt; ++t) full += row_ptr[t + 1] - row_ptr[t] == 64 * nwaves / lpr;
            fprintf(stderr, "[tiles] %s: %d tiles (%d lanes per row, %d window buffers), %.1f rows and %.1f window rows per tile on average, %d full tiles, largest window %d\n",
                    what, nt, lpr, (nbuf == 3 && lpr == 2 && wmax <= 400) ? 3 : 2, (double)row_ptr[nt] / nt, (double)W.win_ptr[nt] / nt, full, wmax);
            fprintf(stderr, "[tiles] %s: %lld nonzeros in %lld slots (%.3f filled), %d of %zu slices stream entries\n", what,
                    (long long)bulk[0]->ptr.back(), (long long)H.sptr.back(), (double)bulk[0]->ptr.back() / (double)std::max(1, H.sptr.back()),
                    over, H.sptr.size() - 1);
        }
    }
    const std::vector<unsigned> zero(16, 0u);
    T.counters.upload(zero.data(), zero.size(), st);
    HIP_CHECK(hipStreamSynchronize(st));
    T.dev.ntiles = nt;
    T.dev.wmax = wmax;
    T.dev.lpr = lpr;
    T.dev.nwaves = nwaves;
    T.dev.nbuf = (nbuf == 3 && lpr == 2 && wmax <= 400) ? 3 : 2;
    T.dev.row_ptr = T.row_ptr.p;
    T.dev.win_ptr = T.win_ptr.p;
    T.dev.win_cols = T.win_cols.p;
    T.dev.counters = T.counters.p;
    return true;
}
// ... of a level operator; planes in the level's numbering
static void build_level_tiles(LevelOp &L, const std::vector<CsrZ> &planes, const std::vector<int> &slot_plane, const std::vector<int> &row_ptr,
                              hipStream_t st, int lpr = 2, int nbuf = 2, int nwaves = 8) {
    TileStore &T = L.tiles;
    T = TileStore();
    if (L.groups.empty() || !L.groups[0].is_real || L.groups[0].nplanes != 2) return;   // the tile kernel's bulk group: two real planes
    const size_t ng = L.groups.size();
    {
        const GroupHost &G = L.groups[0];
        std::vector<const CsrZ *> mats;
        for (int q = 0; q < G.nplanes; ++q) mats.push_back(&planes[slot_plane[G.plane0 + q]]);
        if (!build_tiles_core(T, mats, union_pattern(planes), row_ptr, lpr, st, "operator", nbuf, nwaves)) return;
    }
    T.all_symmetric = true;
    for (size_t g = 0; g < ng; ++g) T.all_symmetric = T.all_symmetric && L.groups[g].symmetric;
    // side rows: every entry of the other groups, row by row (level numbering), plane slot and complex value per entry.  transposed:
    // the entries of the groups' transposes (symmetric groups as they are); rows longer than the long-row limit keep an empty CSR row
    // and go to the long list instead
    struct SideHost {
        std::vector<int> of_row, ptr, col, slot, ls_ptr, ls_col, ls_slot, ls_side;
        std::vector<cplx> val, ls_val;
        int nside = 0;
    };
    auto build_side = [&](bool transposed) {
        SideHost S;
        const int64_t n = L.n;
        const int limit = getenv("WAE_LONG_ROW") ? std::max(1, atoi(getenv("WAE_LONG_ROW"))) : WAE_LONG_ROW;
        std::vector<CsrZ> trs;                                 // transposes of the non-symmetric planes, in (group, plane) order
        std::vector<const CsrZ *> src;                         // per (group >= 1, plane): the matrix to take rows from
        std::vector<int> src_slot;
        for (size_t g = 1; g < ng; ++g)
            for (int q = 0; q < L.groups[g].nplanes; ++q) {
                const CsrZ &A = planes[slot_plane[L.groups[g].plane0 + q]];
                src_slot.push_back(L.groups[g].plane0 + q);
                if (transposed && !L.groups[g].symmetric) trs.push_back(csr_transpose(A));
            }
        size_t it = 0;
        for (size_t g = 1; g < ng; ++g)
            for (int q = 0; q < L.groups[g].nplanes; ++q)
                src.push_back(transposed && !L.groups[g].symmetric ? &trs[it++] : &planes[slot_plane[L.groups[g].plane0 + q]]);
        std::vector<int> count((size_t)n, 0);
        for (const CsrZ *A : src)
            for (int64_t i = 0; i < n; ++i) count[(size_t)i] += A->ptr[i + 1] - A->ptr[i];
        S.of_row.assign((size_t)n, -1);
        S.ptr.assign(1, 0);
        S.ls_ptr.assign(1, 0);
        std::vector<char> is_long((size_t)n, 0);
        for (int64_t i = 0; i < n; ++i)
            if (count[(size_t)i]) {
                S.of_row[(size_t)i] = (int)S.ptr.size() - 1;
                is_long[(size_t)i] = transposed && count[(size_t)i] > limit;
                S.ptr.push_back(S.ptr.back() + (is_long[(size_t)i] ? 0 : count[(size_t)i]));
            }
        S.nside = (int)S.ptr.size() - 1;
        S.col.resize((size_t)S.ptr.back()); S.slot.resize((size_t)S.ptr.back()); S.val.resize((size_t)S.ptr.back());
        std::vector<int> fill(S.ptr.begin(), S.ptr.end() - 1);
        for (int64_t i = 0; i < n; ++i) {                      // (long rows: one list per row, entries in (plane, column) order)
            if (!is_long[(size_t)i]) continue;
            for (size_t k = 0; k < src.size(); ++k)
                for (int p = src[k]->ptr[i]; p < src[k]->ptr[i + 1]; ++p) {
                    S.ls_col.push_back(src[k]->col[p]); S.ls_slot.push_back(src_slot[k]);
                    S.ls_val.push_back(cplx{src[k]->val[p].real(), src[k]->val[p].imag()});
                }
            S.ls_ptr.push_back((int)S.ls_col.size());
            S.ls_side.push_back(S.of_row[(size_t)i]);
        }
        for (size_t k = 0; k < src.size(); ++k) {
            const CsrZ &A = *src[k];
            for (int64_t i = 0; i < n; ++i) {
                if (is_long[(size_t)i]) continue;
                for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
                    const int e = fill[(size_t)S.of_row[(size_t)i]]++;
                    S.col[(size_t)e] = A.col[p]; S.slot[(size_t)e] = src_slot[k]; S.val[(size_t)e] = cplx{A.val[p].real(), A.val[p].imag()};
                }
            }
        }
        return S;
    };
    {
        const SideHost S = build_side(false);
        T.side_of_row.upload(S.of_row.data(), S.of_row.size(), st);
        T.side_ptr.upload(S.ptr.data(), S.ptr.size(), st);
        if (S.nside) {
            T.side_col.upload(S.col.data(), S.col.size(), st);
            T.side_slot.upload(S.slot.data(), S.slot.size(), st);
            T.side_val.upload(S.val.data(), S.val.size(), st);
            T.side_acc.alloc((size_t)S.nside * 256);         // batch widths up to 256 columns
        }
        HIP_CHECK(hipStreamSynchronize(st));
        T.dev.nside = S.nside;
        T.dev.side_of_row = T.side_of_row.p;
        T.dev.side_ptr = T.side_ptr.p; T.dev.side_col = T.side_col.p; T.dev.side_slot = T.side_slot.p;
        T.dev.side_val = T.side_val.p; T.dev.side_acc = T.side_acc.p;
        T.dev.nlong_side = 0;
        if (getenv("WAE_SETUP_DEBUG"))
            fprintf(stderr, "[tiles] %d side rows with %d entries of the other %zu groups\n", S.nside, S.ptr.back(), ng - 1);
    }
    // The transposed orientation (op = T / C on a family with a non-symmetric term -- the flame term of the adjoint solves): the bulk
    // group must be symmetric (the tile storage itself is shared), the side rows are those of the other groups' transposes.
    static const bool tile_t_on = !(getenv("WAE_TILE_TRANSPOSED") && atoi(getenv("WAE_TILE_TRANSPOSED")) == 0);
    if (!T.all_symmetric && L.groups[0].symmetric && tile_t_on) {
        const SideHost S = build_side(true);
        T.t_side_of_row.upload(S.of_row.data(), S.of_row.size(), st);
        T.t_side_ptr.upload(S.ptr.data(), S.ptr.size(), st);
        if (S.nside) {
            T.t_side_col.upload(S.col.data(), S.col.size(), st);
            T.t_side_slot.upload(S.slot.data(), S.slot.size(), st);
            T.t_side_val.upload(S.val.data(), S.val.size(), st);
            T.t_side_acc.alloc((size_t)S.nside * 256);
        }
        const int nls = (int)S.ls_side.size();
        if (nls) {
            T.t_ls_ptr.upload(S.ls_ptr.data(), S.ls_ptr.size(), st);
            T.t_ls_col.upload(S.ls_col.data(), S.ls_col.size(), st);
            T.t_ls_slot.upload(S.ls_slot.data(), S.ls_slot.size(), st);
            T.t_ls_val.upload(S.ls_val.data(), S.ls_val.size(), st);
            T.t_ls_side.upload(S.ls_side.data(), S.ls_side.size(), st);
            T.t_ls_part.alloc((size_t)nls * WAE_LONG_SPLIT * 256);
        }
        HIP_CHECK(hipStreamSynchronize(st));
        T.dev_t = T.dev;
        T.dev_t.nside = S.nside;
        T.dev_t.side_of_row = T.t_side_of_row.p;
        T.dev_t.side_ptr = T.t_side_ptr.p; T.dev_t.side_col = T.t_side_col.p; T.dev_t.side_slot = T.t_side_slot.p;
        T.dev_t.side_val = T.t_side_val.p; T.dev_t.side_acc = T.t_side_acc.p;
        T.dev_t.nlong_side = nls;
        T.dev_t.ls_ptr = T.t_ls_ptr.p; T.dev_t.ls_col = T.t_ls_col.p; T.dev_t.ls_slot = T.t_ls_slot.p; T.dev_t.ls_val = T.t_ls_val.p;
        T.dev_t.ls_side = T.t_ls_side.p;
        T.dev_t.ls_part = T.t_ls_part.p;
        T.ready_t = true;
        if (getenv("WAE_SETUP_DEBUG"))
            fprintf(stderr, "[tiles] transposed orientation: %d side rows with %d entries, %d long rows with %d entries\n", S.nside, S.ptr.back(), nls,
                    S.ls_ptr.back());
    }
    T.ready = true;
}

// ----------------------------------------------------------------------------------------------------
// input conversion
// ----------------------------------------------------------------------------------------------------
// body(lo, hi) over contiguous ranges of [0, n) on up to nth host threads; an exception of any range is rethrown here
template <class F> static void host_ranges(int64_t n, int nth, F &&body) {
    nth = (int)std::max<int64_t>(1, std::min<int64_t>(nth, n / 65536 + 1));
    if (nth == 1) { body((int64_t)0, n); return; }
    std::vector<std::future<void>> jobs;
    for (int t = 0; t < nth; ++t) {
        const int64_t lo = n * t / nth, hi = n * (t + 1) / nth;
        jobs.push_back(std::async(std::launch::async, [&body, lo, hi]() { body(lo, hi); }));
    }
    std::exception_ptr first;
    for (auto &j : jobs) {
        try { j.get(); } catch (...) { if (!first) first = std::current_exception(); }
    }
    if (first) std::rethrow_exception(first);
}

static CsrZ term_to_csr(int64_t d, int index_bytes, int base, int orientation, const void *ptr, const void *idx, const double *val) {
    auto getp = [&](int64_t i) -> int64_t { return index_bytes == 4 ? (int64_t)((const uint32_t *)ptr)[i] : ((const int64_t *)ptr)[i]; };
    auto geti = [&](int64_t i) -> int64_t { return index_bytes == 4 ? (int64_t)((const uint32_t *)idx)[i] : ((const int64_t *)idx)[i]; };
    CsrZ A;
    A.n = A.m = d;
    const int64_t nnz = getp(d) - base;
    WAE_REQUIRE(nnz >= 0 && nnz < (int64_t)2147483647, "term nnz out of range");
    A.ptr.resize(d + 1);
    A.col.resize(nnz);
    A.val.resize(nnz);
    // (four threads per term, the terms themselves side by side in wae_family_create_opts: copying and checking 30 M entries of a
    // 1M-unknown family on one thread was 1.0 s of the 1.7 s a family takes to create)
    constexpr int NTH = 4;
    host_ranges(d + 1, NTH, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t p = getp(i) - base;
            WAE_REQUIRE(p >= 0 && p <= nnz, "pointer array out of range");
            A.ptr[i] = (int)p;
        }
    });
    host_ranges(nnz, NTH, [&](int64_t lo, int64_t hi) {
        for (int64_t p = lo; p < hi; ++p) {
            const int64_t j = geti(p) - base;
            WAE_REQUIRE(j >= 0 && j < d, "index out of range");
            A.col[p] = (int)j;
            A.val[p] = zc(val[2 * p], val[2 * p + 1]);
        }
    });
    // rows already sorted without duplicates (what scipy and SparseArrays hand over): taken as they are
    std::atomic<bool> canonical{true};
    host_ranges(d, NTH, [&](int64_t lo, int64_t hi) {
        bool ok = true;
        for (int64_t i = lo; i < hi; ++i) {
            WAE_REQUIRE(A.ptr[i] <= A.ptr[i + 1], "pointer array not monotone");
            for (int p = A.ptr[i] + 1; p < A.ptr[i + 1]; ++p) ok = ok && A.col[p - 1] < A.col[p];
        }
        if (!ok) canonical = false;
    });
    if (canonical) return orientation == WAE_CSC ? csr_transpose(A) : A;
    // sort + merge duplicates per row
    CsrZ S;
    S.n = S.m = d;
    S.ptr.assign(d + 1, 0);
    std::vector<std::pair<int, zc>> row;
    for (int64_t i = 0; i < d; ++i) {
        row.clear();
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) row.emplace_back(A.col[p], A.val[p]);
        std::stable_sort(row.begin(), row.end(), [](const std::pair<int, zc> &a, const std::pair<int, zc> &b) { return a.first < b.first; });
        for (size_t k = 0; k < row.size(); ++k) {
            if (!S.col.empty() && (int)S.col.size() > S.ptr[i] && S.col.back() == row[k].first) S.val.back() += row[k].second;
            else { S.col.push_back(row[k].first); S.val.push_back(row[k].second); }
        }
        S.ptr[i + 1] = (int)S.col.size();
    }
    if (orientation == WAE_CSC) return csr_transpose(S);
    return S;
}

// term k == s * plane q exactly?
static bool proportional(const CsrZ &A, const CsrZ &P, zc &s) {
    if (!csr_same_pattern(A, P) || A.nnz() == 0) return false;
    int64_t p0 = -1;
    for (int64_t p = 0; p < P.nnz(); ++p)
        if (P.val[p] != zc(0)) { p0 = p; break; }
    if (p0 < 0) return false;
    s = A.val[p0] / P.val[p0];
    for (int64_t p = 0; p < P.nnz(); ++p)
        if (A.val[p] != s * P.val[p]) return false;
    return true;
}

// ... of the restriction R (rows: the coarse level's tile numbering, columns: the fine level's): consecutive rows are cut into tiles
// whose fine-level window fits LDS
static void build_restriction_tiles(Transfer &X, const CsrD &R, int wcap, hipStream_t st) {
    X.r_tiles = TileStore();
    CsrZ Rz, Zz;                                             // plane 0 = R, plane 1 = 0 (the kernel's bulk group has two planes)
    Rz.n = R.n; Rz.m = R.m; Rz.ptr = R.ptr; Rz.col = R.col;
    Rz.val.resize(R.val.size());
    for (size_t i = 0; i < R.val.size(); ++i) Rz.val[i] = zc(R.val[i], 0.0);
    Zz.n = R.n; Zz.m = R.m; Zz.ptr = R.ptr; Zz.col = R.col;
    Zz.val.assign(R.val.size(), zc(0.0, 0.0));
    Pattern U;
    U.n = R.n; U.ptr = R.ptr; U.col = R.col;
    std::vector<int> row_ptr(1, 0), stamp((size_t)R.m, -1);
    int rows = 0, win = 0;
    for (int64_t i = 0; i < R.n; ++i) {
        if (R.ptr[i + 1] - R.ptr[i] > wcap) return;          // (a row that does not fit a window)
        int fresh = 0;
        const int t = (int)row_ptr.size() - 1;
        for (int p = R.ptr[i]; p < R.ptr[i + 1]; ++p) fresh += stamp[(size_t)R.col[p]] != t;
        if (rows == 128 || win + fresh > wcap) {
            row_ptr.push_back((int)i);
            rows = 0; win = 0;
        }
        const int t2 = (int)row_ptr.size() - 1;
        for (int p = R.ptr[i]; p < R.ptr[i + 1]; ++p)
            if (stamp[(size_t)R.col[p]] != t2) { stamp[(size_t)R.col[p]] = t2; ++win; }
        ++rows;
    }
    row_ptr.push_back((int)R.n);
    if (!build_tiles_core(X.r_tiles, {&Rz, &Zz}, U, row_ptr, 4, st, "restriction")) return;
    X.r_tiles.dev.unit = 1;
    X.r_tiles.ready = true;
}

// The prolongation by fine tile (wae_internal.h XferTiles): P (fine x coarse, rows in the fine level's tile order), row_ptr = the fine tiles.
static void build_transfer_tiles(Transfer &X, const CsrD &P, const std::vector<int> &row_ptr, hipStream_t st) {
    XferTiles &F = X.ft;
    F.ready = false;
    const int nt = (int)row_ptr.size() - 1;
    if (nt <= 0 || row_ptr.back() != P.n) return;
    std::vector<int> tptr(nt + 1, 0), clist, stamp((size_t)P.m, -1), slot((size_t)P.m, 0);
    std::vector<unsigned short> ploc(P.col.size());
    int maxslots = 0, maxent = 0;
    std::vector<int> cols;
    for (int t = 0; t < nt; ++t) {
        const int a = row_ptr[t], b = row_ptr[t + 1];
        if (b - a > 256) return;                             // (the kernel walks at most 256 fine rows per workgroup)
        cols.clear();
        for (int p = P.ptr[a]; p < P.ptr[b]; ++p)
            if (stamp[(size_t)P.col[p]] != t) { stamp[(size_t)P.col[p]] = t; cols.push_back(P.col[p]); }
        std::sort(cols.begin(), cols.end());
        const int ns = (int)cols.size();
        maxslots = std::max(maxslots, ns);
        maxent = std::max(maxent, P.ptr[b] - P.ptr[a]);
        for (int k = 0; k < ns; ++k) slot[(size_t)cols[k]] = k;
        clist.insert(clist.end(), cols.begin(), cols.end());
        tptr[t + 1] = (int)clist.size();
        for (int p = P.ptr[a]; p < P.ptr[b]; ++p) ploc[p] = (unsigned short)slot[(size_t)P.col[p]];
    }
    maxent = (maxent + 3) & ~3;
    if ((size_t)maxslots * 128 + (size_t)maxent * 10 + 1100 > 60 * 1024) return;       // (LDS of the kernel)
    F.row_ptr.upload(row_ptr.data(), row_ptr.size(), st);
    F.tptr.upload(tptr.data(), tptr.size(), st);
    F.clist.upload(clist.data(), clist.size(), st);
    F.pptr.upload(P.ptr.data(), P.ptr.size(), st);
    F.ploc.upload(ploc.data(), ploc.size(), st);
    F.pval.upload(P.val.data(), P.val.size(), st);
    HIP_CHECK(hipStreamSynchronize(st));                     // (the host vectors die at scope end)
    XferTilesDev &D = F.dev;
    D.ntiles = nt; D.maxslots = maxslots; D.maxent = maxent; D.nslots = (int64_t)clist.size(); D.nf = P.n; D.nc = P.m;
    D.row_ptr = F.row_ptr.p; D.tptr = F.tptr.p; D.clist = F.clist.p; D.pptr = F.pptr.p; D.ploc = F.ploc.p; D.pval = F.pval.p;
    F.ready = true;
}
// WAE_XFER_TILES=0: the older prolongation kernel (A/B measurements, tests)
static bool xfer_tiles_on() { const char *e = getenv("WAE_XFER_TILES"); return !(e && atoi(e) == 0); }

// ----------------------------------------------------------------------------------------------------
// coefficient tables
// ----------------------------------------------------------------------------------------------------
// plane coefficients for one system from term coefficients (aliased terms folded in), conj for op = C
static void plane_coeffs(const wae_family *h, const double *coeffs, int op, std::vector<zc> &pc) {
    pc.assign(h->nplanes, zc(0));
    for (int k = 0; k < h->T; ++k) pc[h->term_plane[k]] += h->term_scale[k] * zc(coeffs[2 * k], coeffs[2 * k + 1]);
    if (op == WAE_OP_C)
        for (auto &c : pc) c = std::conj(c);
}
// upload [level][sys][slot] tables
static void upload_pc(wae_family *h, const std::vector<std::vector<zc>> &pcs) {
    const int nsys = (int)pcs.size();
    const int nl = (int)h->ops.size();
    const size_t per_level = (size_t)nsys * h->nplanes;
    std::vector<cplx> tab(per_level * (nl + 2));
    for (int l = 0; l <= nl + 1; ++l) {                  // blocks nl, nl+1: the penalty block and the penalty rows (own slot orders)
        if (l >= nl && h->n_penalty == 0) break;
        const std::vector<int> &sp = l < nl ? h->slot_plane[l] : (l == nl ? h->pen_slot : h->pen_row_slot);
        for (int s = 0; s < nsys; ++s)
            for (int q = 0; q < h->nplanes; ++q) {
                const zc c = pcs[s][sp[q]];
                tab[l * per_level + (size_t)s * h->nplanes + q] = cplx{c.real(), c.imag()};
            }
    }
    h->pc_stride_level = per_level;
    h->pcdev.upload(tab.data(), tab.size(), h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
}
static inline const cplx *pc_level(const wae_family *h, int l) { return h->pcdev.p + (size_t)l * h->pc_stride_level; }

// ----------------------------------------------------------------------------------------------------
// multigrid V-cycle (all columns in lock-step)
// ----------------------------------------------------------------------------------------------------
struct Batch {
    int nb;     // columns (leading dimension of every multivector)
    int cps;    // columns per system
    int nsys;
    int op;
};

static void dense_setup(wae_family *h, const Batch &bt) {
    const int L = (int)h->ops.size() - 1;
    if (h->nc <= 0) return;
    HIP_CHECK(hipMemsetAsync(h->dstatus.p, 0, sizeof(int), h->stream));
    launch_dense_assemble(h->dense_planes.p, h->nplanes, (int)h->nc, pc_level(h, L), bt.nsys, bt.op, h->Ainv.p, h->stream);
    launch_dense_invert(h->Ainv.p, (int)h->nc, bt.nsys, h->dstatus.p, h->stream);
    int st = 0;
    HIP_CHECK(hipMemcpyAsync(&st, h->dstatus.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    if (st) throw WaeError(WAE_ERR_BREAKDOWN, "coarse operator is singular");
}

// weight of the PRE-smoothing sweeps: the set-up's (opts[2]) -- or, in the light cycle of the projected phase, WAE_JAC_LIGHT (experiments)
static double pre_weight(const wae_family *h) {
    static const double w_light = getenv("WAE_JAC_LIGHT") ? atof(getenv("WAE_JAC_LIGHT")) : 0.0;
    return h->vc_light ? (w_light > 0.0 ? w_light : h->jac_w_light) : h->jac_w;
}
// x = Minv b on level l;  returns pointer to the result (either lx[l] or lt[l])
// have_x0: the first sweep of level l (x = w/diag b) is already in lx[l] (written by the SpMV that produced b, MODE_AX_J0)
// final_out (level 0 only): the last post-smoothing sweep writes its result there (a Krylov basis slot) instead of into lx/lt
static cplx *vcycle(wae_family *h, const Batch &bt, int l, const cplx *b, const unsigned char *cm = nullptr, bool have_x0 = false,
                    cplx *final_out = nullptr) {
    const int L = (int)h->ops.size() - 1;
    hipStream_t st = h->stream;
    if (l == L) {
        launch_dense_apply(h->Ainv.p, (int)h->nc, bt.cps, b, h->lx[l].p, bt.nb, st, cm);
        return h->lx[l].p;
    }
    const OpDev A = h->ops[l].dev(bt.op);
    const cplx *pc = pc_level(h, l);
    cplx *x = h->lx[l].p, *t = h->lt[l].p;
    if (!have_x0) launch_jacobi0(A, pc, bt.cps, b, x, pre_weight(h), bt.nb, st, cm);
    for (int s = 1; s < h->nsweeps; ++s) {
        launch_spmv(A, pc, bt.cps, x, t, b, pre_weight(h), bt.nb, MODE_JAC, st, cm);
        std::swap(x, t);
    }
    // (experiment) WAE_VC_LIGHT_ADD=1: the light cycle additive -- the coarse correction is computed from b itself, not from the residual
    // after the first sweep: no second fine-level product per cycle
    static const int light_add = getenv("WAE_VC_LIGHT_ADD") ? atoi(getenv("WAE_VC_LIGHT_ADD")) : 0;
    // residual -> t, restrict -> lb[l+1]
    const bool additive = h->vc_light && light_add && l == 0 && h->nsweeps == 1;
    if (!additive) launch_spmv(A, pc, bt.cps, x, t, b, 0.0, bt.nb, MODE_RES, st, cm);
    // for op = T/C the transfer operators are unchanged (real): (R A P)^H = R A^H P
    const bool by_tile = h->xfer[l].ft.ready && bt.nb >= 8 && xfer_tiles_on();      // (wae_internal.h XferTiles: level 0, wide batches)
    launch_spmv(h->xfer[l].devR(), h->one_dev.p, 1 << 30, additive ? b : t, h->lb[l + 1].p, nullptr, 0.0, bt.nb, MODE_AX, st, cm);
    const cplx *xc = vcycle(h, bt, l + 1, h->lb[l + 1].p, cm);
    if (by_tile) launch_prolong_tiles(h->xfer[l].ft.dev, xc, x, bt.nb, st, cm);
    else launch_prolong_add(h->xfer[l].p_ptr.p, h->xfer[l].p_col.p, h->xfer[l].p_val.p, h->xfer[l].nf, xc, x, bt.nb, st, cm);
    // Post-smoothing on the coarse levels (WAE_VC_POST_COARSE): 1 always, 0 never, 2 (default) everywhere but in the projected phase of
    // a contour integral.  A solve that starts from a projected guess (216 of the 256 points of the benchmark contour) takes 1-5 steps:
    // there a cheaper cycle beats a better one (measured at 1M unknowns: projected phase 1.17 -> 0.99 s without the coarse
    // post-smoothing, while the from-zero solves of the snapshot phase lose 1.26 -> 1.63 s).
    static const int post_coarse = getenv("WAE_VC_POST_COARSE") ? atoi(getenv("WAE_VC_POST_COARSE")) : 2;
    // ... and the fine level's too (WAE_VC_LIGHT_POST0=1 keeps it): the light cycle is V(1,0) on every level -- projected phase 1.02 -> 0.89 s,
    // same eigenpairs (residuals 5.6e-9 -> 6.4e-9, rank gap 1.2e9), 9 325 -> 9 139 column-iterations per pass.
    static const int light_post0 = getenv("WAE_VC_LIGHT_POST0") ? atoi(getenv("WAE_VC_LIGHT_POST0")) : 0;
    const bool no_post = l >= 1 ? (post_coarse == 0 || (post_coarse == 2 && h->vc_light)) : (h->vc_light && !light_post0);
    const int npost = no_post ? 0 : h->nsweeps;
    // (WAE_JAC_POST: a post-smoothing weight of its own -- two sweeps with different weights form a degree-2 polynomial smoother)
    static const double w_post_env = getenv("WAE_JAC_POST") ? atof(getenv("WAE_JAC_POST")) : 0.0;
    const double w_post = w_post_env > 0.0 ? w_post_env : h->jac_w_post;
    for (int s = 0; s < npost; ++s) {
        cplx *dst = (final_out && s == npost - 1) ? final_out : t;
        launch_spmv(A, pc, bt.cps, x, dst, b, w_post, bt.nb, MODE_JAC, st, cm);
        if (dst == t) std::swap(x, t); else x = dst;
    }
    if (final_out && x != final_out) { launch_copy(x, final_out, (size_t)h->ops[l].n * bt.nb, st); x = final_out; }
    return x;
}

// ----------------------------------------------------------------------------------------------------
// batched right-preconditioned GMRES(m)
// ----------------------------------------------------------------------------------------------------
struct ColState {
    std::vector<zc> H;      // (m+1) x m column-major upper part after rotations
    std::vector<zc> Hraw;   // the same columns before the rotations (pair steps of the narrow batches)
    std::vector<zc> g;
    std::vector<double> cs;
    std::vector<zc> sn;
    std::vector<zc> cdef;   // deflation: c_j = u^H M^-1 A v_j, the component removed from every new Krylov vector
    int steps = 0;          // Arnoldi steps to use for the update
    bool conv = false;
};

// A solution that starts from a guess assembled out of other solutions (wae_beyn_moments_rb) is accurate in the norm of
// the stopping test but not componentwise on the penalty rows: their unknowns are ~1e-11 of the rest and are multiplied by
// 1e15 in the operator, so eigenvectors built from such solutions show a large residual exactly there.  (From a zero guess
// the ~30 V-cycle applications of the Krylov process resolve them as a by-product.)  This solves the penalty rows' own
// equations  A_bb d = (b - A x)_b  for the given interior values: the block is an admittance-scaled boundary mass matrix,
// for which point relaxation with weight 0.8 contracts by 0.6 per sweep (spectrum of D^-1 M_P1,2D in [1/2, 2]); the sweeps
// run on the compact n_b x n_b operator and cost microseconds.  Best effort: a block of another kind on which the
// relaxation does not contract is left as it was.
static void penalty_polish(wae_family *h, const Batch &bt, const cplx *B, cplx *X) {
    if (h->n_penalty <= 0) return;
    // Sweeps: 22 damped point relaxations whose weights are the reciprocals of the Chebyshev nodes of [0.4, 2.2] -- the interval that
    // holds the spectrum of D^-1 A_bb for a P1 boundary mass matrix ([1/2, 2]) with a margin -- taken in an order that keeps the partial
    // products bounded (round 4: the same 1e-9 as 40 sweeps with the fixed weight 0.8, whose contraction is 0.6 per sweep; 1.3 -> 0.7 ms
    // per chunk of the projected phase).  WAE_PEN_SWEEPS=<n> restores n fixed-weight sweeps (0: no polish).
    static const int sweeps_fixed = getenv("WAE_PEN_SWEEPS") ? atoi(getenv("WAE_PEN_SWEEPS")) : -1;
    if (sweeps_fixed == 0) return;
    static const std::vector<double> cheb = []() {
        const int n = 22;
        const double lo = 0.4, hi = 2.2, th = 0.5 * (hi + lo), de = 0.5 * (hi - lo);
        std::vector<double> w(n);
        for (int k = 0; k < n; ++k) w[k] = 1.0 / (th - de * std::cos((2 * k + 1) * M_PI / (2 * n)));
        std::vector<double> o;                                    // nodes from both ends inwards: large and small weights alternate
        for (int a = 0, b = n - 1; a <= b; ++a, --b) { o.push_back(w[a]); if (a != b) o.push_back(w[b]); }
        return o;
    }();
    const int sweeps = sweeps_fixed > 0 ? sweeps_fixed : (int)cheb.size();
    auto weight = [&](int s_) { return sweeps_fixed > 0 ? 0.8 : cheb[(size_t)s_]; };
    hipStream_t st = h->stream;
    const int nb = bt.nb;
    const int64_t nbk = h->n_penalty;
    const OpDev A = h->ops[0].dev(bt.op);
    const OpDev Ab = h->pen_op.dev(bt.op);
    const cplx *pcb = pc_level(h, (int)h->ops.size());
    if (bt.op == WAE_OP_N && h->pen_row_op.n == nbk) {
        // residual on the penalty rows only: their rows of the operator as a small (n_b x d) operator of its own -- the full
        // fine-level SpMV this replaces cost as much as a Krylov step's operator product per chunk
        launch_gather_rows(B, h->pen_rows.p, nbk, nb, h->pen_t.p, st);
        launch_spmv(h->pen_row_op.dev(WAE_OP_N), pc_level(h, (int)h->ops.size() + 1), bt.cps, X, h->pen_b.p, h->pen_t.p, 0.0, nb, MODE_RES, st);
    } else {
        launch_spmv(A, pc_level(h, 0), bt.cps, X, h->W.p, B, 0.0, nb, MODE_RES, st);
        launch_gather_rows(h->W.p, h->pen_rows.p, nbk, nb, h->pen_b.p, st);
    }
    launch_norms(h->pen_b.p, nbk, nb, h->partial.p, h->hdev.p, st);
    cplx *x = h->pen_x.p, *t = h->pen_t.p;
    launch_jacobi0(Ab, pcb, bt.cps, h->pen_b.p, x, weight(0), nb, st);
    for (int s = 1; s < sweeps; ++s) {
        launch_spmv(Ab, pcb, bt.cps, x, t, h->pen_b.p, weight(s), nb, MODE_JAC, st);
        std::swap(x, t);
    }
    launch_spmv(Ab, pcb, bt.cps, x, t, h->pen_b.p, 0.0, nb, MODE_RES, st);
    launch_norms(t, nbk, nb, h->partial.p, h->hdev.p + nb, st);
    cplx *hp = h->h_pinned;
    HIP_CHECK(hipMemcpyAsync(hp, h->hdev.p, (size_t)2 * nb * sizeof(cplx), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    for (int b = 0; b < nb; ++b)
        if (!(hp[nb + b].x <= 1e-3 * hp[b].x)) return;          // not contracting (or NaN): leave X alone
    launch_scatter_add_rows(x, h->pen_rows.p, nbk, nb, X, st);
}

// ----------------------------------------------------------------------------------------------------
// wide-batch GMRES with the recurrence on the device
// ----------------------------------------------------------------------------------------------------
// The same left-preconditioned, lock-step, unnormalised-basis GMRES(m) as `gmres` below (its "lazy" branch), with the
// per-column Hessenberg / Givens / convergence bookkeeping in kernels (kernels.hip gmres_*_kernel): an iteration is a chain of
// launches with no device-to-host copy; the host looks at three status words every WAE_GMRES_SYNC iterations (default 4) and at
// the per-column figures once per restart cycle.  Columns that converge between two looks are masked on the device at once
// (their 8-column chunks are skipped by every kernel), so the overshoot costs launches, not traffic.
static double now_s();
static int gmres_wide(wae_family *h, const Batch &bt, const cplx *B, cplx *X, double tol, int maxit, wae_solve_info *info, bool have_x0) {
    hipStream_t st = h->stream;
    const double t_dbg0 = now_s();
    const int nb = bt.nb;
    const int64_t n = h->d;
    const size_t vec = (size_t)n * nb;
    const int m = (int)std::min<size_t>(150, h->V.n / vec - 1);
    const OpDev A = h->ops[0].dev(bt.op);
    const cplx *pc = pc_level(h, 0);
    cplx *hp = h->h_pinned;
    static const int ksync = getenv("WAE_GMRES_SYNC") ? std::max(1, atoi(getenv("WAE_GMRES_SYNC"))) : 4;
    static const double lim = getenv("WAE_LAZY_LIMIT") ? atof(getenv("WAE_LAZY_LIMIT")) : 1e100;
    static const char *env_mask = getenv("WAE_MASK");
    const bool use_mask = env_mask ? atoi(env_mask) != 0 : have_x0;
    const int nch = (nb + 7) / 8;
    const int histcap = maxit + m + 8;
    // device state
    auto ens = [](auto &buf, size_t cnt) { if (buf.n < cnt) buf.alloc(cnt); };
    ens(h->gs_R, (size_t)m * (m + 1) * nb); ens(h->gs_sn, (size_t)m * nb); ens(h->gs_g, (size_t)(m + 1) * nb); ens(h->gs_rescale, (size_t)nb);
    ens(h->gs_cs, (size_t)m * nb); ens(h->gs_sv, (size_t)(m + 2) * nb); ens(h->gs_relres, (size_t)nb); ens(h->gs_bnorm, (size_t)nb);
    ens(h->gs_hist, (size_t)histcap * nb); ens(h->gs_int, (size_t)5 * nb + 4); ens(h->gs_done, (size_t)nb);
    if (h->cmask.n < (size_t)nch) h->cmask.alloc(nch);
    if (h->vsq.n < (size_t)(m + 2) * nb) h->vsq.alloc((size_t)(m + 2) * nb);
    GmresDev S;
    S.nb = nb; S.m = m; S.histcap = histcap;
    S.R = h->gs_R.p; S.cs = h->gs_cs.p; S.sn = h->gs_sn.p; S.g = h->gs_g.p; S.sv = h->gs_sv.p; S.vsq = h->vsq.p;
    S.conv = h->gs_int.p; S.steps = S.conv + nb; S.iters = S.steps + nb; S.histlen = S.iters + nb; S.stalled = S.histlen + nb; S.status = S.stalled + nb;
    S.relres = h->gs_relres.p; S.bnorm = h->gs_bnorm.p; S.hist = h->gs_hist.p; S.rescale = h->gs_rescale.p; S.cmask = h->cmask.p;
    // Pair steps (kernels.hip "Two Arnoldi steps per pass over the basis"): from iteration pair_min of a cycle on, while most columns
    // are still active, the operator is applied twice before the Gram-Schmidt pass.  Same Krylov space, same per-column stopping test
    // after each of the two steps; what it costs is one operator application when the batch ends on the first step of a pair.
    static const int pair_min = getenv("WAE_GMRES_PAIR") ? atoi(getenv("WAE_GMRES_PAIR")) : 2;       // (< 0: off; 4 until round 4: 2.02 -> 1.99 s per pass)
    const bool pair_on = pair_min >= 0 && lim >= 1e50 && nb >= 8 && h->ops.size() > 1;
    ens(h->gs_Hraw, (size_t)m * (m + 1) * nb); ens(h->gs_pair, ((size_t)4 * (m + 3) + 8) * nb);
    if (h->gs_sub.n < (size_t)m * nb) h->gs_sub.alloc((size_t)m * nb);
    S.Hraw = h->gs_Hraw.p; S.sub = h->gs_sub.p;
    cplx *const pr_c1 = h->gs_pair.p, *const pr_c2 = pr_c1 + (size_t)(m + 3) * nb, *const pr_c2m = pr_c2 + (size_t)(m + 3) * nb,
               *const pr_hd2 = pr_c2m + (size_t)(m + 3) * nb, *const pr_gram = pr_hd2 + (size_t)(m + 3) * nb, *const pr_alpha = pr_gram + (size_t)3 * nb,
               *const pr_norm = pr_alpha + nb;
    HIP_CHECK(hipMemsetAsync(h->gs_int.p, 0, ((size_t)5 * nb + 4) * sizeof(int), st));
    if (!have_x0) launch_fill_zero(X, vec, st);
    cplx *const zb = vcycle(h, bt, 0, B);                    // M^-1 b: its norm scales the stopping test; from a zero guess it is also
    launch_norms(zb, n, nb, h->partial.p, h->hdev.p, st);    // the first preconditioned residual (nothing touches the V-cycle's
    HIP_CHECK(hipMemcpyAsync(hp, h->hdev.p, nb * sizeof(cplx), hipMemcpyDeviceToHost, st));   // buffers until then)
    HIP_CHECK(hipStreamSynchronize(st));
    std::vector<double> bnorm(nb), relres(nb, 0.0);
    std::vector<int> iters(nb, 0), hostint((size_t)5 * nb + 4);
    std::vector<unsigned char> done(nb, 0);
    std::vector<char> stalled(nb, 0);
    for (int b = 0; b < nb; ++b) { bnorm[b] = hp[b].x; if (!(bnorm[b] > 0.0)) done[b] = 1; }
    {
        std::vector<double> bn(bnorm);
        for (double &v : bn) if (!(v > 0.0)) v = 1.0;
        HIP_CHECK(hipMemcpyAsync(h->gs_bnorm.p, bn.data(), nb * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    int total_it = 0;
    bool first = !have_x0, x0_unchecked = have_x0, nan_seen = false;
    while (true) {
        cplx *z0;
        if (first) {
            z0 = zb;
        } else {
            launch_spmv(A, pc, bt.cps, X, h->W.p, B, 0.0, nb, MODE_RES, st);
            z0 = vcycle(h, bt, 0, h->W.p);
        }
        first = false;
        launch_norms(z0, n, nb, h->partial.p, h->hdev.p, st);
        HIP_CHECK(hipMemcpyAsync(hp, h->hdev.p, (size_t)nb * sizeof(cplx), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (x0_unchecked) {
            // a guess that is worse than no guess (an ill-conditioned projected system) is dropped, column by column
            x0_unchecked = false;
            std::vector<cplx> keep(nb, cplx{1.0, 0.0});
            bool any_bad = false;
            for (int b = 0; b < nb; ++b)
                if (bnorm[b] > 0.0 && !(hp[b].x / bnorm[b] <= 1.0)) { keep[b] = cplx{0.0, 0.0}; any_bad = true; }
            if (any_bad) {
                h->ydev.upload(keep.data(), nb, st);
                launch_mask_cols(X, h->ydev.p, n, nb, st);
                HIP_CHECK(hipStreamSynchronize(st));
                continue;
            }
        }
        bool all_done = true;
        for (int b = 0; b < nb; ++b) {
            if (bnorm[b] > 0.0) {
                relres[b] = hp[b].x / bnorm[b];
                if (std::isnan(relres[b])) nan_seen = true;
                done[b] = relres[b] <= tol || stalled[b];
            }
            if (!done[b]) all_done = false;
        }
        if (all_done || total_it >= maxit || nan_seen) break;
        launch_scale_inv(z0, h->hdev.p, h->V.p, n, nb, st);                       // V0 = M^-1 r / beta
        HIP_CHECK(hipMemcpyAsync(h->gs_done.p, done.data(), nb, hipMemcpyHostToDevice, st));
        launch_gmres_init(S, h->hdev.p, h->gs_done.p, use_mask ? 1 : 0, st);
        HIP_CHECK(hipStreamSynchronize(st));                                       // (`done` is reused by the host below)
        const unsigned char *mk = (use_mask && nb >= 8) ? h->cmask.p : nullptr;
        int j = 0;
        int status[4] = {0, 0, 0, 0};
        for (int b = 0; b < nb; ++b) status[0] += done[b] ? 0 : 1;              // columns still to converge at the start of this cycle
        int since_sync = 0;
        for (; j < m && total_it < maxit;) {
            const cplx *vj = h->V.p + (size_t)j * vec;
            const int nvj = j + 1;
            const bool fuse0 = h->ops.size() > 1;
            if (pair_on && j >= pair_min && j + 2 <= m && total_it + 2 <= maxit && status[0] > nb / 4) {
                cplx *w1 = h->V.p + (size_t)nvj * vec, *w2 = w1 + vec;         // computed in their basis slots, orthogonalised in place
                launch_spmv(A, pc, bt.cps, vj, h->W.p, h->lx[0].p, pre_weight(h), nb, MODE_AX_J0, st, mk);
                vcycle(h, bt, 0, h->W.p, mk, true, w1);
                launch_spmv(A, pc, bt.cps, w1, h->W.p, h->lx[0].p, pre_weight(h), nb, MODE_AX_J0, st, mk);
                vcycle(h, bt, 0, h->W.p, mk, true, w2);
                launch_dots2_scaled(h->V.p, vec, nvj, w1, w2, n, nb, h->partial.p, pr_c1, pr_c2, pr_gram, h->vsq.p, st, mk);
                launch_gmres_pair_coef(S, j, pr_c1, pr_c2, pr_gram, pr_alpha, pr_c2m, pr_hd2, st);
                launch_axpy2_norm(h->V.p, vec, nvj, pr_c1, pr_c2m, pr_alpha, w1, w2, n, nb, h->partial.p, pr_norm, h->vsq.p + (size_t)nvj * nb, st, mk);
                // (the middle vector is not renormalised in place -- the relation of the vector after it was formed with it as it is;
                // its norm is within one operator application of a vector the range guard has seen)
                launch_gmres_step(S, pr_c1, j, tol, 1e300, use_mask ? 1 : 0, w1, n, st, pr_norm);
                launch_gmres_step(S, pr_hd2, j + 1, tol, lim, use_mask ? 1 : 0, w2, n, st, pr_norm + nb);
                j += 2;
                total_it += 2;
                since_sync += 2;
            } else {
                launch_spmv(A, pc, bt.cps, vj, h->W.p, fuse0 ? h->lx[0].p : nullptr, fuse0 ? pre_weight(h) : 0.0, nb, fuse0 ? MODE_AX_J0 : MODE_AX, st, mk);
                cplx *w = vcycle(h, bt, 0, h->W.p, mk, fuse0);
                launch_dots_scaled(h->V.p, vec, nvj, w, n, nb, h->partial.p, h->hdev.p, h->vsq.p, st, mk);
                launch_axpy_neg_norm(h->V.p, vec, nvj, h->hdev.p, h->V.p + (size_t)nvj * vec, n, nb, h->partial.p, h->hdev.p + (size_t)nvj * nb, st, mk,
                                     w, h->vsq.p + (size_t)nvj * nb);
                launch_gmres_step(S, h->hdev.p, j, tol, lim, use_mask ? 1 : 0, h->V.p + (size_t)nvj * vec, n, st);
                ++j;
                ++total_it;
                ++since_sync;
            }
            // look at the status words every ksync iterations -- every iteration once few columns are left (the end of the
            // cycle is near: an overshoot iteration is ~25 launches of fully masked kernels)
            // (from a projected guess most columns need one or two steps: the first four steps of such a cycle are looked at one by one --
            // a look costs a 12-byte copy and a stream wait, a step run in vain 25 launches of partly masked kernels)
            if (since_sync >= ((have_x0 && j <= 4) ? 1 : ksync) || j == m || total_it >= maxit || status[0] <= nb / 4) {
                since_sync = 0;
                HIP_CHECK(hipMemcpyAsync(status, S.status, 3 * sizeof(int), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                if (status[1]) nan_seen = true;
                if (status[0] == 0 || nan_seen) break;
            }
        }
        // x += V y on the device, then the per-column figures of this cycle
        if (j > 0) {
            launch_gmres_solve_y(S, j, h->ydev.p, st);
            launch_lincomb_add(h->V.p, vec, j, h->ydev.p, X, n, nb, st);
        }
        HIP_CHECK(hipMemcpyAsync(hostint.data(), h->gs_int.p, ((size_t)5 * nb + 4) * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(hp, h->gs_relres.p, nb * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        bool any_stalled_now = false;
        const double *rr = (const double *)hp;
        for (int b = 0; b < nb; ++b) {
            iters[b] = hostint[(size_t)2 * nb + b];
            stalled[b] = (char)hostint[(size_t)4 * nb + b];
            any_stalled_now = any_stalled_now || stalled[b];
            if (hostint[(size_t)nb + b] > 0 && bnorm[b] > 0.0) relres[b] = rr[b];      // columns that took steps in this cycle
        }
        if (hostint[(size_t)5 * nb + 1]) nan_seen = true;
        if (nan_seen) break;
        if (j <= 12 && !any_stalled_now) {
            bool all_est = true;
            for (int b = 0; b < nb; ++b) if (bnorm[b] > 0.0 && !(relres[b] <= tol)) all_est = false;
            if (all_est) break;
        }
    }
    if (have_x0 && !nan_seen) penalty_polish(h, bt, B, X);
    static const bool dbg = getenv("WAE_GMRES_DEBUG") && atoi(getenv("WAE_GMRES_DEBUG"));
    if (dbg) {
        HIP_CHECK(hipStreamSynchronize(st));
        int itot = 0;
        for (int b = 0; b < nb; ++b) itot += iters[b];
        fprintf(stderr, "[gmres] nb=%d x0=%d lockstep_its=%d column_its=%d %.1f ms (device recurrence)\n", nb, (int)have_x0, total_it, itot,
                (now_s() - t_dbg0) * 1e3);
        if (atoi(getenv("WAE_GMRES_DEBUG")) > 2) {                 // per-column step counts, one system per line
            for (int b0 = 0; b0 < nb; b0 += bt.cps) {
                fprintf(stderr, "[gmres]   steps");
                for (int b = b0; b < std::min(nb, b0 + bt.cps); ++b) fprintf(stderr, " %d", iters[b]);
                fprintf(stderr, "\n");
            }
        }
    }
    if (info) {
        int imax = 0, itot = 0, nun = 0;
        double rmax = 0.0;
        for (int b = 0; b < nb; ++b) {
            imax = std::max(imax, iters[b]);
            itot += iters[b];
            if (bnorm[b] > 0.0) {
                if (!(relres[b] <= tol)) ++nun;
                rmax = std::max(rmax, relres[b]);
            }
        }
        info->iters_max = std::max(info->iters_max, imax);
        info->iters_total += itot;
        info->n_unconverged += nun;
        for (int b = 0; b < nb; ++b) if (stalled[b] && !(relres[b] <= tol)) info->levels |= 1 << 16;
        info->relres_max = std::max(info->relres_max, rmax);
        info->levels = (info->levels & (1 << 16)) | (int)h->ops.size();
    }
    if (nan_seen) throw WaeError(WAE_ERR_NAN, "NaN in GMRES");
    return total_it;
}

// returns the number of lock-step iterations
static int gmres(wae_family *h, const Batch &bt, const cplx *B, cplx *X, double tol, int maxit, wae_solve_info *info,
                 const cplx *guess_dir = nullptr, bool have_x0 = false) {
    // LEFT-preconditioned GMRES(m) on  M^-1 A x = M^-1 b  (M^-1 = one multigrid V-cycle), all columns in lock-step.
    // Every norm is therefore a norm of the preconditioned residual M^-1 r ~ the error itself.  This matters here:
    // the admittance rows carry 1e15-sized entries (Helmholtz.jl:151-156), so the plain residual norm is dominated
    // by a handful of boundary rows and says nothing about the interior (a right-preconditioned version accepted
    // x = x0/z as "converged" to 1e-17 in inveriter's first step).
    hipStream_t st = h->stream;
    const double t_dbg0 = now_s();
    const int nb = bt.nb;
    const int64_t n = h->d;
    const size_t vec = (size_t)n * nb;
    // narrow batches get a longer recurrence from the same workspace (near-singular systems in the Newton-type
    // solvers stall under short restarts); they also afford a second Gram-Schmidt pass
    // Deflation of a known near-null direction g (guess_dir; the Newton-type solvers pass their current eigenvector
    // estimate): with u = M^-1 A g, u^ = u/||u||, the Krylov process runs on P M^-1 A, P = I - u^ u^H (u^ sits in front of
    // the basis and takes part in the Gram-Schmidt step; the coefficient c_j = u^H M^-1 A v_j it removes is kept), and
    // the solution is x = V y + alpha g with alpha = (u^H r0 - sum_j y_j c_j)/||u||, which cancels the u^ component of
    // the residual exactly.  Close to an eigenvalue of the NLEVP the operator is nearly singular along g: the undeflated
    // solves needed 50-100 iterations of a long recurrence there, the deflated operator behaves like a regular shift.
    static const bool env_defl = !(getenv("WAE_DEFLATE") && atoi(getenv("WAE_DEFLATE")) == 0);
    // (a single-level hierarchy is the exact dense inverse: nothing to deflate, and the inverse of a numerically singular
    // small matrix is not something to build a projector from)
    static const char *env_re = getenv("WAE_REORTH");
    const bool reorth = env_re ? atoi(env_re) != 0 : nb <= 8;
    // Pair steps in the two-pass recurrence (kernels.hip "Two Arnoldi steps per pass over the basis"): w1 = Op v_j, w2 = Op w1, both
    // orthogonalised against V_0..j by two passes of classical Gram-Schmidt that read the basis ONCE each for the two vectors -- four
    // readings of the basis per two steps instead of eight; same Krylov space, Hessenberg columns recovered on the host (below).
    // For batches whose basis vectors are large enough for the readings to be what an iteration costs (>= 4 MB per vector).
    // With the deflation vector u^ (below): w1 is projected, w1 <- (I - u^ u^H) w1, before the operator is applied to it -- one inner
    // product and one update with a single vector -- so that w2 = Op (P Op v_j) continues the recurrence of P Op; u^ then takes part in
    // the two passes like a basis vector.  WAE_NARROW_PAIR=0: off, =1: at every size.
    static const int narrow_pair = getenv("WAE_NARROW_PAIR") ? atoi(getenv("WAE_NARROW_PAIR")) : -1;
    const bool pair_size = narrow_pair > 0 || (narrow_pair < 0 && vec * sizeof(cplx) >= ((size_t)4 << 20));
    const bool pair_cfg = pair_size && reorth && h->ops.size() > 1 && nb <= 16;
    const bool deflate = guess_dir != nullptr && env_defl && h->ops.size() > 1;
    const int off = deflate ? 1 : 0;
    // (recurrence length of the narrow batches: WAE_GMRES_NARROW_M, default 150 -- the basis buffer of the wide batches holds it)
    static const int narrow_m = getenv("WAE_GMRES_NARROW_M") ? std::max(10, atoi(getenv("WAE_GMRES_NARROW_M"))) : 150;
    const int m = (int)std::min<size_t>((size_t)narrow_m, h->V.n / vec - 1) - off;
    // wide batches, single Gram-Schmidt pass: unnormalised basis (see the inner loop); WAE_LAZY=0 restores the normalisation pass
    static const bool lazy_on = !(getenv("WAE_LAZY") && atoi(getenv("WAE_LAZY")) == 0);
    const size_t nslots = (size_t)m + off + 2;             // basis slots incl. the deflation vector and the newest vector
    const bool lazy = lazy_on && !reorth && nslots * nb <= 4096 && nslots * nb <= h->vsq.n;   // 4096: coefficients of one axpy launch
    static const bool dev_rec = !(getenv("WAE_GMRES_DEVICE") && atoi(getenv("WAE_GMRES_DEVICE")) == 0);
    if (dev_rec && lazy && !guess_dir && nb > 8 && nb <= 256) return gmres_wide(h, bt, B, X, tol, maxit, info, have_x0);
    std::vector<std::vector<double>> sv(lazy ? nslots : 0, std::vector<double>(nb, 1.0));
    const int pair_min = 4;
    const size_t PK = (size_t)(m + off + 3) * nb;             // one coefficient block of the pair steps
    cplx *pr_dev = nullptr, *pr_host = nullptr;
    if (pair_cfg) {
        const size_t need = 6 * PK + 16 * (size_t)nb;
        if (h->gs_pair.n < need) h->gs_pair.alloc(need);
        if (h->h_pin_pair_n < need) {
            if (h->h_pin_pair) { (void)hipHostFree(h->h_pin_pair); h->h_pin_pair = nullptr; h->h_pin_pair_n = 0; }
            HIP_CHECK(hipHostMalloc((void **)&h->h_pin_pair, need * sizeof(cplx)));
            h->h_pin_pair_n = need;
        }
        pr_dev = h->gs_pair.p; pr_host = h->h_pin_pair;
        if (h->vsq.n < PK) h->vsq.alloc(PK);
        std::vector<cplx> ones(PK, cplx{1.0, 0.0});          // the basis is normalised: unit scales for the scaled inner products
        h->vsq.upload(ones.data(), ones.size(), st);
        HIP_CHECK(hipStreamSynchronize(st));
    }
    const OpDev A = h->ops[0].dev(bt.op);
    const cplx *pc = pc_level(h, 0);
    cplx *hp = h->h_pinned;
    // have_x0: X already holds an initial guess (the Galerkin projection on earlier solutions, beyn_moments_rb); the
    // stopping test stays relative to ||M^-1 b||, so the answer is the same as from a zero guess, only cheaper
    if (!have_x0) launch_fill_zero(X, vec, st);
    cplx *const zb0 = vcycle(h, bt, 0, B);                   // M^-1 b: scales the stopping test; from a zero guess (and without a guess
    launch_norms(zb0, n, nb, h->partial.p, h->hdev.p, st);   // direction, whose set-up runs further V-cycles) also the first residual
    HIP_CHECK(hipMemcpyAsync(hp, h->hdev.p, nb * sizeof(cplx), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    std::vector<double> bnorm(nb), relres(nb, 0.0);
    std::vector<int> iters(nb, 0);
    std::vector<char> done(nb, 0), stalled(nb, 0);
    std::vector<std::vector<double>> hist(nb);
    // Attainable accuracy of the residual itself.  Close to an eigenvalue of the NLEVP the solution is ~1/mu times the right-hand side and
    // M^-1 amplifies along the same direction, so a recomputed M^-1 (b - A x) carries rounding of the order eps |A| |x| ||M^-1|| while
    // the recurrence's estimate has reached the tolerance (with an accurate deflation direction the large part of x stays out of the
    // residual, see add_alpha_g below; with a poor one -- the first left solve of a Newton step -- it does not: 1e-2..1e-3 of ||M^-1 b||
    // at 1M DoF).  What is left is a multiple of the near-null direction, which a further cycle cannot remove and an inverse-iteration
    // step does not care about.  A column whose recomputed residual is > 50 x the estimate its cycle ended with (estimate <= tol)
    // switches the batch to cycles of 5 steps (enough to show whether a fresh recurrence still gains: the clean rate is ~0.55 per step),
    // and from then on EVERY column has to halve its recomputed residual per cycle or ends as stalled -- whichever column raised the
    // flag: the columns of a batch sit at their floors one after the other, and with the test tied to the flagged column only the
    // right solves of a Newton step at 1M DoF spent 30 of their 78 steps on a residual that stayed at 2.6e-10.
    std::vector<double> drift_ref(nb, 0.0);         // the recomputed residual at the previous cycle start
    bool drift_mode = false;
    for (int b = 0; b < nb; ++b) { bnorm[b] = hp[b].x; if (!(bnorm[b] > 0.0)) done[b] = 1; }
    std::vector<ColState> cs(nb);
    int total_it = 0;
    bool first = !have_x0;
    bool x0_unchecked = have_x0;
    bool nan_seen = false;
    // converged-chunk mask: columns are skipped in groups of 8 (one 128-B segment of every interleaved row) as soon as
    // all 8 have converged -- the columns of one shifted system converge together, so this removes most of the work the
    // lock-step batch would otherwise spend on finished systems
    const int nch = (nb + 7) / 8;
    // from a zero guess all systems of a batch converge within a few iterations of each other and the predicates cost
    // ~2 %: off unless WAE_MASK=1.  With projected initial guesses the columns start 0..8 digits from the answer and
    // finish at very different times: on (measured -15 % on the C2 Beyn pass), WAE_MASK=0 disables.
    static const char *env_mask = getenv("WAE_MASK");
    const bool use_mask = env_mask ? atoi(env_mask) != 0 : have_x0;
    std::vector<unsigned char> cm(nch, 1), cm_prev(nch, 2);
    if (h->cmask.n < (size_t)nch) h->cmask.alloc(nch);
    auto push_mask = [&]() {
        if (cm != cm_prev) {
            HIP_CHECK(hipMemcpyAsync(h->cmask.p, cm.data(), nch, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipStreamSynchronize(st));
            cm_prev = cm;
        }
    };
    const unsigned char *mk = nullptr;
    std::vector<double> unorm(nb, 0.0);
    std::vector<zc> beta0(nb, zc(0));
    if (deflate) {
        launch_spmv(A, pc, bt.cps, guess_dir, h->W.p, nullptr, 0.0, nb, MODE_AX, st);
        launch_copy(vcycle(h, bt, 0, h->W.p), h->V.p, vec, st);
        launch_norms(h->V.p, n, nb, h->partial.p, h->hdev.p, st);
        launch_norms(guess_dir, n, nb, h->partial.p, h->hdev.p + nb, st);
        HIP_CHECK(hipMemcpyAsync(hp, h->hdev.p, (size_t)2 * nb * sizeof(cplx), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        // ||M^-1 A g|| <= 1e-12 ||g||: g is a null vector to rounding, u^ would be noise -- no deflation for that column
        std::vector<cplx> un(nb);
        for (int b = 0; b < nb; ++b) {
            unorm[b] = (hp[b].x > 1e-12 * hp[nb + b].x && hp[nb + b].x > 0.0) ? hp[b].x : 0.0;
            un[b] = cplx{unorm[b], 0.0};
        }
        h->ydev.upload(un.data(), nb, st);
        launch_scale_inv(h->V.p, h->ydev.p, h->V.p, n, nb, st);          // a column without deflation gets u^ = 0
        HIP_CHECK(hipStreamSynchronize(st));
    }
    if (guess_dir && !deflate) {
        // initial guess x0 = alpha * g, alpha = (M^-1 A g)^H (M^-1 b) / ||M^-1 A g||^2 per column: when the solution is
        // dominated by a known direction (inverse iteration close to an eigenvalue) the Krylov solve only has to
        // produce the small rest
        launch_spmv(A, pc, bt.cps, guess_dir, h->W.p, nullptr, 0.0, nb, MODE_AX, st);
        launch_copy(vcycle(h, bt, 0, h->W.p), h->U.p, vec, st);
        const cplx *zb = vcycle(h, bt, 0, B);
        launch_dots(h->U.p, 0, 1, zb, n, nb, h->partial.p, h->hdev.p, st);
        launch_dots(h->U.p, 0, 1, h->U.p, n, nb, h->partial.p, h->hdev.p + nb, st);
        HIP_CHECK(hipMemcpyAsync(hp, h->hdev.p, (size_t)2 * nb * sizeof(cplx), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        std::vector<cplx> al(nb);
        for (int b = 0; b < nb; ++b) {
            const double den = hp[nb + b].x;
            al[b] = den > 0.0 ? cplx{hp[b].x / den, hp[b].y / den} : cplx{0.0, 0.0};
        }
        h->ydev.upload(al.data(), nb, st);
        launch_lincomb(guess_dir, 0, 1, h->ydev.p, X, n, nb, st);
        HIP_CHECK(hipStreamSynchronize(st));
        first = false;
    }
    while (true) {
        cplx *z0;
        if (first) {
            z0 = (guess_dir == nullptr) ? zb0 : vcycle(h, bt, 0, B);       // (with a guess direction the buffers have been used again)
        } else {
            launch_spmv(A, pc, bt.cps, X, h->W.p, B, 0.0, nb, MODE_RES, st);
            z0 = vcycle(h, bt, 0, h->W.p);
        }
        first = false;
        if (deflate) {                                   // r0 <- P r0, beta0 = u^H r0
            launch_dots(h->V.p, 0, 1, z0, n, nb, h->partial.p, h->hdev.p + nb, st);
            launch_axpy_neg(h->V.p, 0, 1, h->hdev.p + nb, z0, n, nb, st);
        }
        launch_norms(z0, n, nb, h->partial.p, h->hdev.p, st);
        HIP_CHECK(hipMemcpyAsync(hp, h->hdev.p, (size_t)2 * nb * sizeof(cplx), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (deflate) for (int b = 0; b < nb; ++b) beta0[b] = zc(hp[nb + b].x, hp[nb + b].y);
        if (x0_unchecked) {
            // a guess that is worse than no guess (an ill-conditioned projected system) is dropped, column by column
            x0_unchecked = false;
            std::vector<cplx> keep(nb, cplx{1.0, 0.0});
            bool any_bad = false;
            for (int b = 0; b < nb; ++b)
                if (bnorm[b] > 0.0 && !(hp[b].x / bnorm[b] <= 1.0)) { keep[b] = cplx{0.0, 0.0}; any_bad = true; }
            if (any_bad) {
                h->ydev.upload(keep.data(), nb, st);
                launch_mask_cols(X, h->ydev.p, n, nb, st);
                HIP_CHECK(hipStreamSynchronize(st));
                continue;
            }
        }
        bool all_done = true;
        for (int b = 0; b < nb; ++b) {
            if (bnorm[b] > 0.0) {
                relres[b] = hp[b].x / bnorm[b];
                if (std::isnan(relres[b])) nan_seen = true;
                done[b] = relres[b] <= tol || stalled[b];
                if (!done[b]) {
                    const double est = hist[b].empty() ? -1.0 : hist[b].back();
                    if (drift_mode && drift_ref[b] > 0.0 && relres[b] > 0.5 * drift_ref[b]) { stalled[b] = 1; done[b] = 1; }   // a 5-step cycle lies behind this column
                    else if (est >= 0.0 && est <= tol && relres[b] > 50.0 * est) drift_mode = true;
                }
                drift_ref[b] = relres[b];
            }
            if (!done[b]) all_done = false;
        }
        if (getenv("WAE_GMRES_DEBUG") && atoi(getenv("WAE_GMRES_DEBUG")) > 1) {
            double rmax = 0.0, emax = 0.0;
            for (int b = 0; b < nb; ++b) { rmax = std::max(rmax, relres[b]); if (!hist[b].empty()) emax = std::max(emax, hist[b].back()); }
            fprintf(stderr, "[gmres]   cycle start at %d steps: true relres max %.2e (last estimate %.2e)\n", total_it, rmax, emax);
        }
        if (all_done || total_it >= maxit || nan_seen) {
            // the projected residual is small, but its u^ component (beta0) has not been cancelled yet for THIS residual:
            // x += (beta0/||u||) g.  (When the right-hand side lies along M^-1 A g -- an Arnoldi step started from an
            // eigenvector -- that is the whole solution.)
            if (deflate && !nan_seen) {
                std::vector<cplx> al(nb, cplx{0.0, 0.0});
                for (int b = 0; b < nb; ++b) {
                    if (!(unorm[b] > 0.0)) continue;
                    const zc a = beta0[b] / unorm[b];
                    if (std::isfinite(a.real()) && std::isfinite(a.imag())) al[b] = cplx{a.real(), a.imag()};
                }
                h->ydev.upload(al.data(), nb, st);
                launch_lincomb(guess_dir, 0, 1, h->ydev.p, h->U.p, n, nb, st);
                launch_add(h->U.p, X, vec, st);
                HIP_CHECK(hipStreamSynchronize(st));
            }
            break;
        }
        launch_scale_inv(z0, h->hdev.p, h->V.p + (size_t)off * vec, n, nb, st);     // V0 = M^-1 r / beta
        if (lazy) {                                      // slots 0..off hold unit vectors
            for (int i = 0; i <= off; ++i) std::fill(sv[i].begin(), sv[i].end(), 1.0);
            std::vector<cplx> ones((size_t)(off + 1) * nb, cplx{1.0, 0.0});
            h->vsq.upload(ones.data(), ones.size(), st);
            HIP_CHECK(hipStreamSynchronize(st));
        }
        if (use_mask && nb >= 8) {
            for (int k = 0; k < nch; ++k) cm[k] = 0;
            for (int b = 0; b < nb; ++b) if (!done[b]) cm[b >> 3] = 1;
            push_mask();
            mk = h->cmask.p;
        }
        for (int b = 0; b < nb; ++b) {
            ColState &c = cs[b];
            c.H.assign((size_t)(m + 1) * m, zc(0));
            if (pair_cfg) c.Hraw.assign((size_t)(m + 1) * m, zc(0));
            c.g.assign(m + 1, zc(0));
            c.g[0] = hp[b].x;
            c.cs.assign(m, 0.0);
            c.sn.assign(m, zc(0));
            c.cdef.assign(m, zc(0));
            c.steps = 0;
            c.conv = done[b];
        }
        // one new column (raw[0..jc+1]) of the Hessenberg matrix of batch column b: rotations, residual estimate, stopping tests
        auto absorb = [&](int b, int jc, const zc *raw, zc cdef_j) {
            ColState &c = cs[b];
            if (c.conv) return;
            zc *Hc = &c.H[(size_t)jc * (m + 1)];
            for (int i = 0; i <= jc + 1; ++i) Hc[i] = raw[i];
            if (pair_cfg) std::copy(raw, raw + jc + 2, &c.Hraw[(size_t)jc * (m + 1)]);
            if (deflate) c.cdef[jc] = cdef_j;
            for (int i = 0; i < jc; ++i) {
                const zc a = Hc[i], bb = Hc[i + 1];
                Hc[i] = c.cs[i] * a + c.sn[i] * bb;
                Hc[i + 1] = -std::conj(c.sn[i]) * a + c.cs[i] * bb;
            }
            const zc a = Hc[jc];
            const double bb = Hc[jc + 1].real();
            const double aa = std::abs(a);
            const double t = std::sqrt(aa * aa + bb * bb);
            if (!(t > 0.0) || std::isnan(t)) { c.conv = true; if (std::isnan(t)) nan_seen = true; return; }
            if (aa == 0.0) { c.cs[jc] = 0.0; c.sn[jc] = 1.0; }
            else { c.cs[jc] = aa / t; c.sn[jc] = (a / aa) * (bb / t); }
            Hc[jc] = c.cs[jc] * a + c.sn[jc] * bb;
            Hc[jc + 1] = 0;
            c.g[jc + 1] = -std::conj(c.sn[jc]) * c.g[jc];
            c.g[jc] = c.cs[jc] * c.g[jc];
            c.steps = jc + 1;
            iters[b]++;
            relres[b] = std::abs(c.g[jc + 1]) / bnorm[b];
            hist[b].push_back(relres[b]);
            const size_t hs = hist[b].size();
            if (relres[b] <= 0.7 * tol) c.conv = true;
            else if (hs > 60 && relres[b] > 0.9 * hist[b][hs - 31]) { c.conv = true; stalled[b] = 1; }   // attainable accuracy reached
        };
        std::vector<zc> rawcol((size_t)m + 3);
        int j = 0;
        const int m_cycle = drift_mode ? std::min(m, 5) : m;
        while (j < m_cycle && total_it < maxit) {
            const cplx *vj = h->V.p + (size_t)(off + j) * vec;
            if (pair_cfg && j >= pair_min && j + 2 <= m_cycle && total_it + 2 <= maxit) {
                // ---- two steps per reading of the basis (see the comment at pair_cfg) ----
                const int nv = off + j + 1;                          // orthogonalisation set: u^ (when deflating), v_0..v_j
                cplx *w1 = h->V.p + (size_t)nv * vec, *w2 = w1 + vec;
                cplx *c1a = pr_dev, *c2a = c1a + PK, *c1b = c2a + PK, *c2b = c1b + PK, *c2m = c2b + PK, *zero_al = c2m + PK,
                     *gram = zero_al + nb, *alpha = gram + 3 * (size_t)nb, *nrm = alpha + nb, *inv = nrm + 2 * (size_t)nb,
                     *tdef = inv + 2 * (size_t)nb;
                launch_spmv(A, pc, bt.cps, vj, h->W.p, h->lx[0].p, pre_weight(h), nb, MODE_AX_J0, st, mk);
                vcycle(h, bt, 0, h->W.p, mk, true, w1);
                if (deflate) {                                       // w1 <- P w1, t = u^H w1 kept for the deflation coefficient
                    launch_dots(h->V.p, vec, 1, w1, n, nb, h->partial.p, tdef, st, mk);
                    launch_axpy_neg(h->V.p, vec, 1, tdef, w1, n, nb, st, mk);
                }
                launch_spmv(A, pc, bt.cps, w1, h->W.p, h->lx[0].p, pre_weight(h), nb, MODE_AX_J0, st, mk);
                vcycle(h, bt, 0, h->W.p, mk, true, w2);
                // first pass
                launch_dots2_scaled(h->V.p, vec, nv, w1, w2, n, nb, h->partial.p, c1a, c2a, gram, h->vsq.p, st, mk);
                launch_fill_zero(zero_al, nb, st);
                launch_axpy2_norm(h->V.p, vec, nv, c1a, c2a, zero_al, w1, w2, n, nb, h->partial.p, nrm, inv, st, mk);
                // second pass: coefficients, and the Gram entries of the once-orthogonalised pair
                launch_dots2_scaled(h->V.p, vec, nv, w1, w2, n, nb, h->partial.p, c1b, c2b, gram, h->vsq.p, st, mk);
                HIP_CHECK(hipMemcpyAsync(pr_host, pr_dev, (4 * PK) * sizeof(cplx), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipMemcpyAsync(pr_host + 5 * PK + nb, gram, (size_t)3 * nb * sizeof(cplx), hipMemcpyDeviceToHost, st));
                cplx *htdef = pr_host + 5 * PK + 7 * (size_t)nb;
                if (deflate) HIP_CHECK(hipMemcpyAsync(htdef, tdef, (size_t)nb * sizeof(cplx), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                const cplx *hc1a = pr_host, *hc2a = hc1a + PK, *hc1b = hc2a + PK, *hc2b = hc1b + PK, *hgram = pr_host + 5 * PK + nb;
                cplx *hc2m = pr_host + 4 * PK, *halpha = pr_host + 5 * PK + 4 * (size_t)nb;
                auto Z = [](const cplx &v) { return zc(v.x, v.y); };
                std::vector<zc> beta(nb, zc(0));
                for (int b = 0; b < nb; ++b) {
                    // after the second update: w1'' = w1' - V c1b, w2'' = w2' - V c2b; the second vector is made orthogonal to the
                    // first in the same kernel: beta = (w1''^H w2'') / ||w1''||^2, both from the Gram entries of (w1', w2')
                    double g11 = hgram[b].x;
                    zc g12 = Z(hgram[(size_t)nb + b]);
                    for (int i = 0; i < nv; ++i) {
                        const zc p1 = Z(hc1b[(size_t)i * nb + b]), p2 = Z(hc2b[(size_t)i * nb + b]);
                        g11 -= std::norm(p1);
                        g12 -= std::conj(p1) * p2;
                    }
                    beta[b] = g11 > 0.0 ? g12 / g11 : zc(0);
                    if (!std::isfinite(beta[b].real()) || !std::isfinite(beta[b].imag())) beta[b] = zc(0);
                    halpha[b] = cplx{beta[b].real(), beta[b].imag()};
                    for (int i = 0; i < nv; ++i) {
                        const zc q = Z(hc2b[(size_t)i * nb + b]) - beta[b] * Z(hc1b[(size_t)i * nb + b]);
                        hc2m[(size_t)i * nb + b] = cplx{q.real(), q.imag()};
                    }
                }
                HIP_CHECK(hipMemcpyAsync(c2m, hc2m, (size_t)nv * nb * sizeof(cplx), hipMemcpyHostToDevice, st));
                HIP_CHECK(hipMemcpyAsync(alpha, halpha, (size_t)nb * sizeof(cplx), hipMemcpyHostToDevice, st));
                launch_axpy2_norm(h->V.p, vec, nv, c1b, c2m, alpha, w1, w2, n, nb, h->partial.p, nrm, inv, st, mk);
                launch_scale_inv(w1, nrm, w1, n, nb, st, mk);                    // a zero norm leaves the vector as it is (breakdown: below)
                launch_scale_inv(w2, nrm + nb, w2, n, nb, st, mk);
                cplx *hnrm = pr_host + 5 * PK + 5 * (size_t)nb;
                HIP_CHECK(hipMemcpyAsync(hnrm, nrm, (size_t)2 * nb * sizeof(cplx), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                total_it += 2;
                for (int b = 0; b < nb; ++b) {
                    ColState &c = cs[b];
                    if (c.conv) continue;
                    const double a1 = hnrm[b].x, a2 = hnrm[(size_t)nb + b].x;
                    // column j:  Op v_j = u^ (t + e1) + V cc1 + a1 v_{j+1}   (cc: the sums of the two passes; e: their u^ entries)
                    const int nk = j + 1;
                    std::vector<zc> cc1(nk), cc2(nk);
                    for (int i = 0; i < nk; ++i) {
                        cc1[i] = Z(hc1a[(size_t)(off + i) * nb + b]) + Z(hc1b[(size_t)(off + i) * nb + b]);
                        cc2[i] = Z(hc2a[(size_t)(off + i) * nb + b]) + Z(hc2b[(size_t)(off + i) * nb + b]);
                    }
                    const zc e1 = deflate ? Z(hc1a[b]) + Z(hc1b[b]) : zc(0), e2 = deflate ? Z(hc2a[b]) + Z(hc2b[b]) : zc(0);
                    const zc cdef_j = deflate ? Z(htdef[b]) + e1 : zc(0);
                    for (int i = 0; i < nk; ++i) rawcol[i] = cc1[i];
                    rawcol[nk] = a1;
                    absorb(b, j, rawcol.data(), cdef_j);
                    if (c.conv) continue;
                    if (!(a1 > 0.0)) { c.conv = true; continue; }            // invariant subspace: the first step ended the recurrence
                    // column j+1:  Op v_{j+1} = (w2 - Op V cc1) / a1,  Op V cc1 = V_{0..j+1} (H_{0..j-1} cc1[0..j-1]) + cc1[j] w1  (Arnoldi
                    // relation of the earlier columns),  w1 = V cc1 + a1 v_{j+1},  w2 = V cc2 + beta a1 v_{j+1} + a2 v_{j+2}
                    for (int r = 0; r <= j; ++r) {
                        zc d = 0;
                        for (int i = std::max(0, r - 1); i < j; ++i) d += c.Hraw[(size_t)i * (m + 1) + r] * cc1[i];
                        rawcol[r] = (cc2[r] - d - cc1[j] * cc1[r]) / a1;
                    }
                    rawcol[j + 1] = beta[b] - cc1[j];
                    rawcol[j + 2] = a2 / a1;
                    zc cdef_n = 0;
                    if (deflate) {                                   // u^ part of Op v_{j+1}: (e2 - sum_{i<=j} cc1_i cdef_i) / a1
                        cdef_n = e2 - cc1[j] * cdef_j;
                        for (int i = 0; i < j; ++i) cdef_n -= cc1[i] * c.cdef[i];
                        cdef_n /= a1;
                    }
                    absorb(b, j + 1, rawcol.data(), cdef_n);
                }
                j += 2;
            } else {
            const int nvj = off + j + 1;                         // vectors in the orthogonalisation set (u^ first when deflating)
            const bool fuse0 = h->ops.size() > 1;                // A v_j and the V-cycle's first sweep on it in one kernel
            launch_spmv(A, pc, bt.cps, vj, h->W.p, fuse0 ? h->lx[0].p : nullptr, fuse0 ? pre_weight(h) : 0.0, nb, fuse0 ? MODE_AX_J0 : MODE_AX, st, mk);
            cplx *w = vcycle(h, bt, 0, h->W.p, mk, fuse0);       // w = M^-1 A v_j  (lives in a V-cycle buffer)
            if (lazy) {
                // The basis is kept UNNORMALISED (v_i = s_i V^_i, s_i = 1/||V^_i||): M^-1 A is linear, so w^ = M^-1 A V^_j = w/s_j,
                // the update coefficients of w^ against V^_i are s_i^2 (V^_i^H w^) -- s_j cancels -- and the new vector goes
                // straight into its slot; the host rescales what it reads (h_ij = s_j c_i / s_i, h_{j+1,j} = s_j ||w^'||).
                // Saves the normalisation pass (read + write of one multivector) of every iteration.
                launch_dots_scaled(h->V.p, vec, nvj, w, n, nb, h->partial.p, h->hdev.p, h->vsq.p, st, mk);
                launch_axpy_neg_norm(h->V.p, vec, nvj, h->hdev.p, h->V.p + (size_t)nvj * vec, n, nb, h->partial.p, h->hdev.p + (size_t)nvj * nb, st, mk,
                                     w, h->vsq.p + (size_t)nvj * nb);
            } else {
            launch_dots(h->V.p, vec, nvj, w, n, nb, h->partial.p, h->hdev.p, st, mk);
            if (reorth) {   // CGS2: h += V^H w', w' -= V (V^H w')
                launch_axpy_neg(h->V.p, vec, nvj, h->hdev.p, w, n, nb, st, mk);
                cplx *h2 = h->hdev.p + (size_t)(off + m + 2) * nb;
                launch_dots(h->V.p, vec, nvj, w, n, nb, h->partial.p, h2, st, mk);
                launch_axpy_neg_norm(h->V.p, vec, nvj, h2, w, n, nb, h->partial.p, h->hdev.p + (size_t)nvj * nb, st, mk);
                launch_add(h2, h->hdev.p, (size_t)nvj * nb, st);
            } else {        // the update and the norm of its result in one pass
                launch_axpy_neg_norm(h->V.p, vec, nvj, h->hdev.p, w, n, nb, h->partial.p, h->hdev.p + (size_t)nvj * nb, st, mk);
            }
            launch_scale_inv(w, h->hdev.p + (size_t)nvj * nb, h->V.p + (size_t)nvj * vec, n, nb, st, mk);
            }
            HIP_CHECK(hipMemcpyAsync(hp, h->hdev.p, (size_t)(nvj + 1) * nb * sizeof(cplx), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            if (lazy) {                                  // back to the coefficients of the normalised recurrence
                for (int b = 0; b < nb; ++b) {
                    const double sj = sv[nvj - 1][b];
                    for (int i = 0; i < nvj; ++i) {
                        const double f = sv[i][b] > 0.0 ? sj / sv[i][b] : 0.0;
                        hp[(size_t)i * nb + b].x *= f;
                        hp[(size_t)i * nb + b].y *= f;
                    }
                    const double r = hp[(size_t)nvj * nb + b].x;
                    sv[nvj][b] = r > 0.0 ? 1.0 / r : 0.0;
                    hp[(size_t)nvj * nb + b].x = sj * r;
                }
                // the stored norms are the running products of the sub-diagonal entries: long recurrences could carry them
                // out of range (their squares are used) -- normalise this one vector in place and start over from 1
                bool rescale = false;
                static const double lim = getenv("WAE_LAZY_LIMIT") ? atof(getenv("WAE_LAZY_LIMIT")) : 1e100;   // (small values exercise this branch in tests)
                for (int b = 0; b < nb; ++b) rescale = rescale || sv[nvj][b] > lim || (sv[nvj][b] > 0.0 && sv[nvj][b] < 1.0 / lim);
                if (rescale) {
                    launch_scale_inv(h->V.p + (size_t)nvj * vec, h->hdev.p + (size_t)nvj * nb, h->V.p + (size_t)nvj * vec, n, nb, st, mk);
                    std::vector<cplx> ones(nb, cplx{1.0, 0.0});
                    for (int b = 0; b < nb; ++b) if (sv[nvj][b] > 0.0) sv[nvj][b] = 1.0; else ones[b] = cplx{0.0, 0.0};
                    HIP_CHECK(hipMemcpyAsync(h->vsq.p + (size_t)nvj * nb, ones.data(), (size_t)nb * sizeof(cplx), hipMemcpyHostToDevice, st));
                    HIP_CHECK(hipStreamSynchronize(st));
                }
            }
            ++total_it;
            for (int b = 0; b < nb; ++b) {
                if (cs[b].conv) continue;
                for (int i = 0; i <= j + 1; ++i) rawcol[i] = zc(hp[(size_t)(off + i) * nb + b].x, hp[(size_t)(off + i) * nb + b].y);
                absorb(b, j, rawcol.data(), deflate ? zc(hp[b].x, hp[b].y) : zc(0));
            }
            ++j;
            }
            bool all_conv = true;
            for (int b = 0; b < nb; ++b) all_conv = all_conv && cs[b].conv;
            if (all_conv || nan_seen) break;
            if (mk) {
                for (int k = 0; k < nch; ++k) cm[k] = 0;
                for (int b = 0; b < nb; ++b) if (!cs[b].conv) cm[b >> 3] = 1;
                push_mask();
            }
        }
        mk = nullptr;
        bool any_stalled_now = false;
        for (int b = 0; b < nb; ++b) any_stalled_now = any_stalled_now || stalled[b];
        // y = R^{-1} g per column, zero-padded to j steps;  x += V y
        const int ju = std::min(j, m);
        std::vector<cplx> y((size_t)std::max(ju, 1) * nb, cplx{0.0, 0.0});
        for (int b = 0; b < nb; ++b) {
            ColState &c = cs[b];
            const int k = c.steps;
            std::vector<zc> yy(k);
            for (int i = k - 1; i >= 0; --i) {
                zc s = c.g[i];
                for (int q = i + 1; q < k; ++q) s -= c.H[(size_t)q * (m + 1) + i] * yy[q];
                const zc dgi = c.H[(size_t)i * (m + 1) + i];
                yy[i] = (dgi != zc(0)) ? s / dgi : zc(0);
            }
            for (int i = 0; i < k; ++i) {
                const double f = lazy ? sv[off + i][b] : 1.0;                // x += sum_i y_i s_i V^_i
                y[(size_t)i * nb + b] = cplx{f * yy[i].real(), f * yy[i].imag()};
            }
        }
        if (ju > 0) {
            h->ydev.upload(y.data(), (size_t)ju * nb, st);
            launch_lincomb_add(h->V.p + (size_t)off * vec, vec, ju, h->ydev.p, X, n, nb, st);      // x += V y in one pass over x
            HIP_CHECK(hipStreamSynchronize(st));       // y is a stack vector
        }
        // The multiple alpha g that cancels the u^ component of the residual is NOT added between cycles: close to an eigenvalue it
        // is ~1/mu times the rest of x, and a residual recomputed from x + alpha g carries the rounding of that cancellation
        // (1e-3..1e-4 of ||M^-1 b|| at 1M DoF, where the recurrence's estimate stood at 1e-12: every solve spent a second and third
        // cycle on it and ended "stalled").  x holds the Krylov part only; the loop top projects the u^ component out of its residual
        // and the exits add alpha g once (there from beta0 of that residual, here from the recurrence).
        auto add_alpha_g = [&]() {
            std::vector<cplx> al(nb, cplx{0.0, 0.0});
            for (int b = 0; b < nb; ++b) {
                if (!(unorm[b] > 0.0)) continue;
                zc acc = beta0[b];
                for (int i = 0; i < cs[b].steps; ++i) acc -= zc(y[(size_t)i * nb + b].x, y[(size_t)i * nb + b].y) * cs[b].cdef[i];
                acc /= unorm[b];
                if (std::isfinite(acc.real()) && std::isfinite(acc.imag())) al[b] = cplx{acc.real(), acc.imag()};
            }
            h->ydev.upload(al.data(), nb, st);
            launch_lincomb(guess_dir, 0, 1, h->ydev.p, h->U.p, n, nb, st);
            launch_add(h->U.p, X, vec, st);
            HIP_CHECK(hipStreamSynchronize(st));
        };
        if (nan_seen) break;
        // A short recurrence that ended with every column converged by its Arnoldi estimate needs no confirmation by a
        // true residual (another SpMV + V-cycle): over <= 12 steps the estimate equals the preconditioned residual to
        // rounding.  Long recurrences (Gram-Schmidt drift) and stalled columns are always re-checked at the loop top.
        if (j <= 12 && !any_stalled_now) {
            bool all_est = true;
            for (int b = 0; b < nb; ++b) if (bnorm[b] > 0.0 && !(relres[b] <= tol)) all_est = false;
            if (all_est) { if (deflate) add_alpha_g(); break; }
        }
    }
    if (have_x0 && !nan_seen) penalty_polish(h, bt, B, X);
    static const bool dbg = getenv("WAE_GMRES_DEBUG") && atoi(getenv("WAE_GMRES_DEBUG"));
    if (dbg) {
        double r0max = 0.0;
        for (int b = 0; b < nb; ++b) if (!hist[b].empty()) r0max = std::max(r0max, hist[b][0]);
        HIP_CHECK(hipStreamSynchronize(st));
        fprintf(stderr, "[gmres] nb=%d x0=%d lockstep_its=%d first-step relres max=%.2e tol=%.1e %s%.1f ms\n", nb, (int)have_x0, total_it, r0max, tol,
                pair_cfg ? "pair " : (deflate ? "deflated " : ""), (now_s() - t_dbg0) * 1e3);
    }
    if (info) {
        int imax = 0, itot = 0, nun = 0;
        double rmax = 0.0;
        for (int b = 0; b < nb; ++b) {
            imax = std::max(imax, iters[b]);
            itot += iters[b];
            if (bnorm[b] > 0.0) {
                if (!(relres[b] <= tol)) ++nun;
                rmax = std::max(rmax, relres[b]);
            }
        }
        info->iters_max = std::max(info->iters_max, imax);
        info->iters_total += itot;
        info->n_unconverged += nun;
        for (int b = 0; b < nb; ++b) if (stalled[b] && !(relres[b] <= tol)) info->levels |= 1 << 16;   // stagnation marker (masked off below)
        info->relres_max = std::max(info->relres_max, rmax);
        info->levels = (info->levels & (1 << 16)) | (int)h->ops.size();
    }
    if (nan_seen) throw WaeError(WAE_ERR_NAN, "NaN in GMRES");
    return total_it;
}

// ----------------------------------------------------------------------------------------------------
// helpers
// ----------------------------------------------------------------------------------------------------
static void require_solver(const wae_family *h) {
    if (!h->solver_ready) throw WaeError(WAE_ERR_INVALID, "wae_solver_setup has not been called");
}
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F> static int guarded(F &&f) {
    try {
        return f();
    } catch (const WaeError &e) {
        wae_set_error(e.what());
        return e.code;
    } catch (const std::bad_alloc &) {
        wae_set_error("out of host memory");
        return WAE_ERR_INVALID;
    } catch (const std::exception &e) {
        wae_set_error(e.what());
        return WAE_ERR_INVALID;
    }
}

static int info_code(wae_solve_info &i) {   // also clears the internal stagnation marker bit

    const bool stag = (i.levels & (1 << 16)) != 0;
    i.levels &= 0xFFFF;
    if (i.n_unconverged > 0) return stag ? WAE_WARN_STAGNATION : WAE_WARN_MAXITER;
    return WAE_OK;
}

// ----------------------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------------------
// snapshot basis for projected initial guesses (wae_beyn_moments_rb)
// ----------------------------------------------------------------------------------------------------
// Gaussian elimination with partial pivoting on a small dense complex system (column-major n x n), in place.
// Unknowns whose pivot vanishes are set to zero (a deficient direction of the snapshot basis).
static void small_solve(std::vector<zc> &A, std::vector<zc> &b, int n) {
    std::vector<char> dead(n, 0);
    double amax = 0.0;
    for (const zc &a : A) amax = std::max(amax, std::abs(a));
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = 0.0;
        for (int i = k; i < n; ++i) { const double v = std::abs(A[(size_t)k * n + i]); if (v > best) { best = v; p = i; } }
        if (!(best > 1e-14 * amax)) { dead[k] = 1; continue; }
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(A[(size_t)j * n + k], A[(size_t)j * n + p]);
            std::swap(b[k], b[p]);
        }
        const zc inv = 1.0 / A[(size_t)k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const zc f = A[(size_t)k * n + i] * inv;
            if (f == zc(0)) continue;
            for (int j = k + 1; j < n; ++j) A[(size_t)j * n + i] -= f * A[(size_t)j * n + k];
            b[i] -= f * b[k];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        if (dead[k]) { b[k] = 0; continue; }
        zc sres = b[k];
        for (int j = k + 1; j < n; ++j) sres -= A[(size_t)j * n + k] * b[j];
        b[k] = sres / A[(size_t)k * n + k];
    }
}

static void rb_d2h(wae_family *h, const cplx *src, cplx *dst, size_t cnt) {
    HIP_CHECK(hipMemcpyAsync(dst, src, cnt * sizeof(cplx), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
}

// start an empty basis on the store Q (cap snapshots of d x l); kact = terms with a non-zero coefficient in `table`
static void rb_reset(wae_family *h, cplx *Q, int cap, int l, const double *table, int npts, const cplx *Vinter) {
    RbState &R = h->rb;
    const int T = h->T;
    R.Q = Q; R.cap = cap; R.l = l; R.S = 0;
    R.kact.clear();
    for (int k = 0; k < T; ++k) {
        bool used = false;
        for (int p = 0; p < npts && !used; ++p) used = table[((size_t)p * T + k) * 2] != 0.0 || table[((size_t)p * T + k) * 2 + 1] != 0.0;
        if (used) R.kact.push_back(k);
    }
    const size_t vecl = (size_t)h->d * l;
    R.wait_w();
    if (R.W.n < R.kact.size() * (size_t)cap * vecl) {
        const size_t need = R.kact.size() * (size_t)cap * vecl;
        const int dev = h->device;
        RbState *Rp = &R;
        R.w_job = std::async(std::launch::async, [Rp, need, dev]() {
            HIP_CHECK(hipSetDevice(dev));
            Rp->W.alloc(need);
        });
    }
    R.Hk.assign(R.kact.size() * (size_t)cap * cap * l, zc(0));
    R.g.assign((size_t)cap * l, zc(0));
    if (R.Vi.n < vecl) R.Vi.alloc(vecl);
    HIP_CHECK(hipMemcpyAsync(R.Vi.p, Vinter, vecl * sizeof(cplx), hipMemcpyDeviceToDevice, h->stream));
    R.vi_valid = true;
    if (R.hb.n < (size_t)4 * (cap + 4) * l) R.hb.alloc((size_t)4 * (cap + 4) * l);
    if (R.alpha.n < (size_t)l) R.alpha.alloc(l);
    if (R.alpha2.n < (size_t)cap * l) R.alpha2.alloc((size_t)cap * l);
}

// the store slots S .. S+count-1 hold new raw vectors: orthonormalise them per column against the basis (classical
// Gram-Schmidt, two passes), then extend  g = Q^H V  and every projected term  H_k = Q^H A_k Q  by the new rows/columns.
// W_k = A_k Q stays resident (HBM is plentiful: C2 6.5 GB, C3 16 GB), so a new row costs dot products only.  The new
// vectors are handled four at a time (dots_multi reads the basis once per block): the build is a tall-skinny Gram product.
static void rb_append_block(wae_family *h, int cnt, const std::vector<std::vector<std::vector<zc>>> &pck) {
    RbState &R = h->rb;
    hipStream_t st = h->stream;
    const int64_t d = h->d;
    const int l = R.l, cap = R.cap, S = R.S;
    const size_t vecl = (size_t)d * l;
    std::vector<cplx> hh((size_t)(S + cnt) * cnt * l), n0((size_t)cnt * l), n1(l);
    const OpDev A0 = h->ops[0].dev(WAE_OP_N);
    cplx *qn = R.Q + (size_t)S * vecl;                       // the block of new vectors
    for (int j = 0; j < cnt; ++j) launch_norms(qn + (size_t)j * vecl, d, l, h->partial.p, R.hb.p + (size_t)j * l, st);
    rb_d2h(h, R.hb.p, n0.data(), (size_t)cnt * l);
    // (1) against the existing basis: block classical Gram-Schmidt, two passes
    // (the update of all new vectors in one reading of the basis, with the coefficients where dots_multi left them: no round trip
    // through the host; the basis used to be read once per new vector and pass -- 96 ms of a 1M-DoF pass)
    static const bool multi_axpy = !(getenv("WAE_RB_MULTI_AXPY") && atoi(getenv("WAE_RB_MULTI_AXPY")) == 0);
    const bool fits = (size_t)S * cnt * l * sizeof(cplx) <= 60 * 1024;
    for (int pass = 0; pass < 2 && S > 0; ++pass) {
        launch_dots_multi(R.Q, vecl, S, qn, vecl, cnt, d, l, h->partial.p, R.hb.p, st);       // hb[(i*cnt + j)*l + c]
        if (multi_axpy && fits) {
            launch_axpy_neg_multi(R.Q, vecl, S, R.hb.p, qn, vecl, cnt, d, l, st);
            continue;
        }
        rb_d2h(h, R.hb.p, hh.data(), (size_t)S * cnt * l);
        std::vector<cplx> cj((size_t)S * l);
        for (int j = 0; j < cnt; ++j) {
            for (int i = 0; i < S; ++i)
                for (int c = 0; c < l; ++c) cj[(size_t)i * l + c] = hh[((size_t)i * cnt + j) * l + c];
            R.alpha2.upload(cj.data(), cj.size(), st);
            launch_axpy_neg(R.Q, vecl, S, R.alpha2.p, qn + (size_t)j * vecl, d, l, st);
            HIP_CHECK(hipStreamSynchronize(st));             // cj is re-filled for the next vector
        }
    }
    // (2) inside the block, vector by vector; normalise (a vector that adds nothing to a column's span is zeroed there)
    for (int j = 0; j < cnt; ++j) {
        cplx *q = qn + (size_t)j * vecl;
        for (int pass = 0; pass < 2 && j > 0; ++pass) {
            launch_dots(qn, vecl, j, q, d, l, h->partial.p, R.hb.p, st);
            launch_axpy_neg(qn, vecl, j, R.hb.p, q, d, l, st);
        }
        launch_norms(q, d, l, h->partial.p, R.hb.p, st);
        rb_d2h(h, R.hb.p, n1.data(), l);
        for (int c = 0; c < l; ++c) {
            const double nrm0 = n0[(size_t)j * l + c].x;
            n1[c] = (n1[c].x > 1e-9 * nrm0 && nrm0 > 0.0) ? cplx{n1[c].x, 0.0} : cplx{0.0, 0.0};
        }
        R.alpha.upload(n1.data(), l, st);
        launch_scale_inv(q, R.alpha.p, q, d, l, st);
        HIP_CHECK(hipStreamSynchronize(st));
    }
    // (3) g = Q^H V for the new vectors
    launch_dots_multi(qn, vecl, cnt, R.Vi.p, vecl, 1, d, l, h->partial.p, R.hb.p, st);
    rb_d2h(h, R.hb.p, hh.data(), (size_t)cnt * l);
    for (int j = 0; j < cnt; ++j)
        for (int c = 0; c < l; ++c) R.g[(size_t)(S + j) * l + c] = zc(hh[(size_t)j * l + c].x, hh[(size_t)j * l + c].y);
    // (4) projected terms: new columns (all rows) and new rows (old columns)
    for (size_t ki = 0; ki < R.kact.size(); ++ki) {
        R.wait_w();
        cplx *Wk = R.W.p + ki * (size_t)cap * vecl;
        cplx *wn = Wk + (size_t)S * vecl;
        upload_pc(h, pck[ki]);
        for (int j = 0; j < cnt; ++j)
            launch_spmv(A0, pc_level(h, 0), l, qn + (size_t)j * vecl, wn + (size_t)j * vecl, nullptr, 0.0, l, MODE_AX, st);
        zc *H = &R.Hk[ki * (size_t)cap * cap * l];
        launch_dots_multi(R.Q, vecl, S + cnt, wn, vecl, cnt, d, l, h->partial.p, R.hb.p, st);  // q_i^H A_k q_{S+j}, i < S+cnt
        rb_d2h(h, R.hb.p, hh.data(), (size_t)(S + cnt) * cnt * l);
        for (int i = 0; i < S + cnt; ++i)
            for (int j = 0; j < cnt; ++j)
                for (int c = 0; c < l; ++c) {
                    const cplx v = hh[((size_t)i * cnt + j) * l + c];
                    H[((size_t)(S + j) * cap + i) * l + c] = zc(v.x, v.y);
                }
        // new rows (old columns).  A term A_k = s B with B real and symmetric (K, M of a Helmholtz family) projects to s x (a Hermitian
        // matrix): the new rows follow from the new columns, H[S+j, i] = (s / conj s) conj(H[i, S+j]), without reading A_k Q again.
        zc herm_factor(0);
        bool herm = false;
        {
            static const bool herm_on = !(getenv("WAE_RB_HERMITIAN") && atoi(getenv("WAE_RB_HERMITIAN")) == 0);
            const int kterm = R.kact[ki], pl = h->term_plane[kterm];
            const LevelOp &L0 = h->ops[0];
            for (size_t g = 0; g < L0.groups.size() && herm_on && !herm; ++g)
                for (int q = 0; q < L0.groups[g].nplanes; ++q)
                    if (h->slot_plane[0][(size_t)L0.groups[g].plane0 + q] == pl && L0.groups[g].symmetric && L0.groups[g].is_real) {
                        const zc sc = h->term_scale[kterm];
                        if (sc != zc(0)) { herm = true; herm_factor = sc / std::conj(sc); }
                    }
        }
        if (S > 0 && herm) {
            for (int i = 0; i < S; ++i)
                for (int j = 0; j < cnt; ++j)
                    for (int c = 0; c < l; ++c)
                        H[((size_t)i * cap + S + j) * l + c] = herm_factor * std::conj(H[((size_t)(S + j) * cap + i) * l + c]);
        } else if (S > 0) {
            launch_dots_multi(Wk, vecl, S, qn, vecl, cnt, d, l, h->partial.p, R.hb.p, st);     // (A_k q_i)^H q_{S+j} = conj(row S+j), i < S
            rb_d2h(h, R.hb.p, hh.data(), (size_t)S * cnt * l);
            for (int i = 0; i < S; ++i)
                for (int j = 0; j < cnt; ++j)
                    for (int c = 0; c < l; ++c) {
                        const cplx v = hh[((size_t)i * cnt + j) * l + c];
                        H[((size_t)i * cap + S + j) * l + c] = zc(v.x, -v.y);
                    }
        }
    }
    R.S += cnt;
}

static void rb_append(wae_family *h, int count) {
    RbState &R = h->rb;
    WAE_REQUIRE(R.S + count <= R.cap, "snapshot store is full");
    std::vector<std::vector<std::vector<zc>>> pck(R.kact.size());
    for (size_t ki = 0; ki < R.kact.size(); ++ki) {
        std::vector<double> ek((size_t)2 * h->T, 0.0);
        ek[(size_t)2 * R.kact[ki]] = 1.0;
        pck[ki].resize(1);
        plane_coeffs(h, ek.data(), WAE_OP_N, pck[ki][0]);
    }
    for (int done = 0; done < count; done += 4) rb_append_block(h, std::min(4, count - done), pck);
}

// Galerkin guesses of one chunk, Xs[row][sy*l + c] = Q_c (sum_k c_k(z_sy) Q_c^H A_k Q_c)^{-1} Q_c^H v_c, in two steps: the
// S x S solves (host only, reads the projected terms -- safe to run on a helper thread while the device works on the
// previous chunk), and the application of the coefficients on the device.
static std::vector<cplx> rb_guess_coeffs(const wae_family *h, const double *ct_chunk, int ns) {
    const RbState &R = h->rb;
    const int S = R.S, l = R.l, cap = R.cap, T = h->T, nb = ns * l;
    std::vector<cplx> Y((size_t)S * nb);
    std::vector<zc> Hs((size_t)S * S), rhs(S);
    for (int sy = 0; sy < ns; ++sy) {
        const double *ct = ct_chunk + (size_t)sy * 2 * T;
        for (int k = 0; k < T; ++k)
            if ((ct[2 * k] != 0.0 || ct[2 * k + 1] != 0.0) && std::find(R.kact.begin(), R.kact.end(), k) == R.kact.end())
                throw WaeError(WAE_ERR_INVALID, "a term outside the projected set has a non-zero coefficient: rebuild the basis (mode 1)");
        for (int c = 0; c < l; ++c) {
            std::fill(Hs.begin(), Hs.end(), zc(0));
            for (size_t ki = 0; ki < R.kact.size(); ++ki) {
                const zc ck(ct[2 * R.kact[ki]], ct[2 * R.kact[ki] + 1]);
                if (ck == zc(0)) continue;
                const zc *H = &R.Hk[ki * (size_t)cap * cap * l];
                for (int sc = 0; sc < S; ++sc)
                    for (int i = 0; i < S; ++i) Hs[(size_t)sc * S + i] += ck * H[((size_t)sc * cap + i) * l + c];
            }
            for (int i = 0; i < S; ++i) rhs[i] = R.g[(size_t)i * l + c];
            small_solve(Hs, rhs, S);
            for (int i = 0; i < S; ++i) {
                const zc yv = std::isfinite(rhs[i].real()) && std::isfinite(rhs[i].imag()) ? rhs[i] : zc(0);
                Y[(size_t)i * nb + (size_t)sy * l + c] = cplx{yv.real(), yv.imag()};
            }
        }
    }
    return Y;
}
static void rb_apply_guess(wae_family *h, const std::vector<cplx> &Y, int ns, cplx *X) {
    RbState &R = h->rb;
    const int S = (int)(Y.size() / ((size_t)ns * R.l));
    R.ycoef.upload(Y.data(), Y.size(), h->stream);
    launch_lincomb_rep(R.Q, (size_t)h->d * R.l, S, R.ycoef.p, X, h->d, ns * R.l, R.l, h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
}

// ----------------------------------------------------------------------------------------------------
// Kelleher's accelerated ascending-composition generator (same order as perturbation.jl:2-80)
template <class F> static void for_each_partition(int n, F &&f) {
    std::vector<int> a(n + 1, 0);
    int k = 1, y = n - 1;
    while (k != 0) {
        int x = a[k - 1] + 1;
        k -= 1;
        while (2 * x <= y) { a[k] = x; y -= x; k += 1; }
        const int l = k + 1;
        while (x <= y) {
            a[k] = x; a[l] = y;
            f(a.data(), k + 2);
            x += 1; y -= 1;
        }
        a[k] = x + y;
        y = x + y - 1;
        f(a.data(), k + 1);
    }
}

extern "C" {

const char *wae_last_error(void) { return g_last_error.c_str(); }
const char *wae_version(void) { return "waehip 0.1 (gfx950; fused multi-term CSR SpMV, SA-multigrid GMRES)"; }

int wae_device_count(int *n) {
    return guarded([&]() {
        HIP_CHECK(hipGetDeviceCount(n));
        return WAE_OK;
    });
}

int wae_family_create(wae_family **out, int64_t d, int32_t T, int32_t index_bytes, int32_t base, int32_t orientation,
                      const void *const *ptr, const void *const *idx, const double *const *val, int32_t device) {
    return wae_family_create_opts(out, d, T, index_bytes, base, orientation, ptr, idx, val, device, nullptr, 0);
}

int wae_family_create_opts(wae_family **out, int64_t d, int32_t T, int32_t index_bytes, int32_t base, int32_t orientation,
                           const void *const *ptr, const void *const *idx, const double *const *val, int32_t device, const double *opts,
                           int32_t nopts) {
    return guarded([&]() {
        WAE_REQUIRE(out && d > 0 && T > 0 && T <= 64, "bad d/T");
        WAE_REQUIRE(nopts >= 0 && (nopts == 0 || opts), "bad opts");
        const double sym_tol = nopts > 0 ? opts[0] : 0.0;
        WAE_REQUIRE(sym_tol >= 0.0 && sym_tol <= 1e-8, "opts[0] (symmetry tolerance) must lie in [0, 1e-8]");
        WAE_REQUIRE(index_bytes == 4 || index_bytes == 8, "index_bytes must be 4 or 8");
        WAE_REQUIRE(base == 0 || base == 1, "base must be 0 or 1");
        WAE_REQUIRE(d < 2147483647, "d too large for 32-bit indices");
        int ndev = 0;
        HIP_CHECK(hipGetDeviceCount(&ndev));
        WAE_REQUIRE(device >= 0 && device < ndev, "no such HIP device");
        HIP_CHECK(hipSetDevice(device));
        std::unique_ptr<wae_family> h(new wae_family);
        h->device = device;
        h->d = d;
        h->T = T;
        HIP_CHECK(hipStreamCreate(&h->stream));
        h->term_plane.resize(T);
        h->term_scale.resize(T);
        h->term_nnz.resize(T);
        const bool cdbg = getenv("WAE_SETUP_DEBUG") != nullptr;
        double tc = now_s();
        auto clap = [&](const char *what) { if (cdbg) { const double t = now_s(); fprintf(stderr, "[create] %-34s %.3f s\n", what, t - tc); tc = t; } };
        std::vector<std::future<CsrZ>> conv;                     // the terms' conversions side by side
        for (int k = 0; k < T; ++k)
            conv.push_back(std::async(std::launch::async, [&, k]() { return term_to_csr(d, index_bytes, base, orientation, ptr[k], idx[k], val[k]); }));
        std::vector<CsrZ> conv_out(T);
        {
            std::exception_ptr first;
            for (int k = 0; k < T; ++k) {
                try { conv_out[k] = conv[k].get(); } catch (...) { if (!first) first = std::current_exception(); }
            }
            if (first) std::rethrow_exception(first);
        }
        for (int k = 0; k < T; ++k) {
            CsrZ A = std::move(conv_out[k]);
            h->term_nnz[k] = A.nnz();
            bool found = false;
            for (int q = 0; q < (int)h->planes0.size() && !found; ++q) {
                zc s;
                if (proportional(A, h->planes0[q], s)) { h->term_plane[k] = q; h->term_scale[k] = s; found = true; }
            }
            if (!found) {
                h->term_plane[k] = (int)h->planes0.size();
                h->term_scale[k] = 1.0;
                h->planes0.push_back(std::move(A));
            }
        }
        h->nplanes = (int)h->planes0.size();
        clap("terms to CSR, distinct planes");
        // renumber the rows into compact tiles (tiles.h); WAE_REORDER=0 keeps the caller's numbering (A/B measurements)
        static const bool reorder_on = !(getenv("WAE_REORDER") && atoi(getenv("WAE_REORDER")) == 0);
        if (reorder_on) {
            // fine level: two window buffers of 608 rows x 128 B.  WAE_TILE_NBUF=3: three of 400 (two windows in flight while a
            // third is read) -- measured slower, 966 vs 733 us at 1M unknowns and 64 columns: a chunk costs a wavefront the same
            // ~10 k cycles whether its tile has 174 rows or 256 (lane = row), the gather was not what it waited for.
            const int nbuf0 = getenv("WAE_TILE_NBUF") ? atoi(getenv("WAE_TILE_NBUF")) : 2;          // (read per call: the tests switch it)
            const int wcap = getenv("WAE_TILE_WCAP") ? atoi(getenv("WAE_TILE_WCAP")) : (nbuf0 == 3 ? 400 : 608);
            static const int thick = getenv("WAE_TILE_THICK") ? atoi(getenv("WAE_TILE_THICK")) : 6;
            const double tq0 = now_s();
            TilePlan plan = plan_tiles(union_pattern(h->planes0), 256, wcap, thick);
            const double tq1 = now_s();
            if (!plan.perm.empty()) {
                std::vector<std::future<CsrZ>> jobs;
                for (size_t q = 0; q < h->planes0.size(); ++q)
                    jobs.push_back(std::async(std::launch::async, [&, q]() { return permute_symmetric(h->planes0[q], plan.perm, plan.iperm); }));
                for (size_t q = 0; q < h->planes0.size(); ++q) h->planes0[q] = jobs[q].get();
                h->perm_h = plan.perm;
                h->perm_dev.upload(h->perm_h.data(), h->perm_h.size(), h->stream);
                h->tile_row_ptr = plan.row_ptr;
            }
            if (getenv("WAE_SETUP_DEBUG"))
                fprintf(stderr, "[create] tile plan %.3f s (%zu tiles, largest window %d), permutation of the planes %.3f s\n", tq1 - tq0,
                        plan.row_ptr.empty() ? (size_t)0 : plan.row_ptr.size() - 1, plan.wmax, now_s() - tq1);
        }
        h->ops.resize(1);
        h->slot_plane.resize(1);
        tc = now_s();
        h->slot_plane[0] = build_levelop(h->ops[0], h->planes0, h->stream, sym_tol);
        clap("operator groups (CSR, both orientations)");
        if (!h->tile_row_ptr.empty()) {
            build_level_tiles(h->ops[0], h->planes0, h->slot_plane[0], h->tile_row_ptr, h->stream, 2, getenv("WAE_TILE_NBUF") ? atoi(getenv("WAE_TILE_NBUF")) : 2);
            clap("tile storage");
        }
        cplx one = {1.0, 0.0};
        h->one_dev.upload(&one, 1, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        *out = h.release();
        return WAE_OK;
    });
}

int wae_family_destroy(wae_family *h) {
    return guarded([&]() {
        if (h) {
            (void)hipSetDevice(h->device);
            delete h;
        }
        return WAE_OK;
    });
}

int wae_family_info(const wae_family *h, int64_t *d, int32_t *T, int64_t *nnz_total) {
    return guarded([&]() {
        WAE_REQUIRE(h, "null handle");
        if (d) *d = h->d;
        if (T) *T = h->T;
        if (nnz_total) { int64_t s = 0; for (auto v : h->term_nnz) s += v; *nnz_total = s; }
        return WAE_OK;
    });
}

int64_t wae_family_spmv_bytes(const wae_family *h, const uint8_t *mask, int32_t r) {
    if (!h) return -1;
    int64_t bytes = 0;
    for (int k = 0; k < h->T; ++k)
        if (!mask || mask[k]) bytes += h->term_nnz[k] * 20 + (h->d + 1) * 4;
    return bytes + 2 * (int64_t)r * h->d * 16;
}

static void ensure(DevBuf<cplx> &b, size_t n) { if (b.n < n) b.alloc(n); }

int wae_spmv_sum_cols(wae_family *h, const double *coeffs, int32_t ncoef, const double *X, double *Y, int32_t r, int32_t op) {
    return guarded([&]() {
        WAE_REQUIRE(h && r >= 0 && (r == 0 || (coeffs && X && Y)), "bad argument");
        WAE_REQUIRE(op >= 0 && op <= 2, "bad op");
        if (r == 0) return WAE_OK;                                // L(z) * zeros(d, 0): nothing to do (LinOpFam.jl:482-529 returns d x 0)
        WAE_REQUIRE(ncoef == 1 || ncoef == r, "ncoef must be 1 or r");
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        const size_t cnt = (size_t)h->d * r;
        ensure(h->io_a, cnt); ensure(h->io_b, cnt);
        constexpr int GW = 256;                                   // widest launch (the side-row / long-row scratch is sized for it)
        const int gw = std::min<int>(r, GW);
        DevBuf<cplx> xi, yi;
        xi.alloc((size_t)h->d * gw); yi.alloc((size_t)h->d * gw);
        std::vector<cplx> tab((size_t)ncoef * h->nplanes);
        std::vector<zc> pc;
        for (int s = 0; s < ncoef; ++s) {
            plane_coeffs(h, coeffs + (size_t)s * 2 * h->T, op, pc);
            for (int q = 0; q < h->nplanes; ++q) { const zc c = pc[h->slot_plane[0][q]]; tab[(size_t)s * h->nplanes + q] = cplx{c.real(), c.imag()}; }
        }
        DevBuf<cplx> pcd;
        pcd.upload(tab.data(), tab.size(), st);
        HIP_CHECK(hipMemcpyAsync(h->io_a.p, X, cnt * sizeof(cplx), hipMemcpyHostToDevice, st));
        for (int c0 = 0; c0 < r; c0 += GW) {                      // column groups of at most GW
            const int w = std::min(GW, r - c0);
            launch_colmajor_to_inter(h->io_a.p + (size_t)c0 * h->d, h->d, w, xi.p, w, st, h->perm());
            launch_spmv(h->ops[0].dev(op), ncoef == 1 ? pcd.p : pcd.p + (size_t)c0 * h->nplanes, ncoef == 1 ? (1 << 30) : 1, xi.p, yi.p, nullptr, 0.0,
                        w, MODE_AX, st);
            launch_inter_to_colmajor(yi.p, w, h->d, w, h->io_b.p + (size_t)c0 * h->d, st, h->perm());
        }
        HIP_CHECK(hipMemcpyAsync(Y, h->io_b.p, cnt * sizeof(cplx), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        xi.release(); yi.release(); pcd.release();
        return WAE_OK;
    });
}

int wae_eig_residuals(wae_family *h, int32_t n, const double *coeff_table, const double *P, uint64_t P_dev, double *res_out) {
    if (h && n > 256 && coeff_table && (P || P_dev) && res_out) {  // pairs are independent: groups of at most 256 (the widest launch)
        for (int32_t j0 = 0; j0 < n; j0 += 256) {
            const int32_t w = std::min<int32_t>(256, n - j0);
            const int rc = wae_eig_residuals(h, w, coeff_table + (size_t)j0 * 2 * h->T, P ? P + (size_t)j0 * 2 * h->d : nullptr,
                                             P_dev ? P_dev + (uint64_t)j0 * h->d * sizeof(cplx) : 0, res_out + j0);
            if (rc != WAE_OK) return rc;
        }
        return WAE_OK;
    }
    return guarded([&]() {
        WAE_REQUIRE(h && n >= 0 && (n == 0 || (coeff_table && (P || P_dev) && res_out)), "bad argument");
        if (n == 0) return WAE_OK;                                // no pairs: nothing to test
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        const int T = h->T;
        const size_t cnt = (size_t)h->d * n;
        DevBuf<cplx> xi, yi, pcd, nrm, part;
        xi.alloc(cnt); yi.alloc(cnt); nrm.alloc((size_t)n); part.alloc((size_t)1024 * n);
        const cplx *src = (const cplx *)(uintptr_t)P_dev;
        if (!src) {
            ensure(h->io_a, cnt);
            HIP_CHECK(hipMemcpyAsync(h->io_a.p, P, cnt * sizeof(cplx), hipMemcpyHostToDevice, st));
            src = h->io_a.p;
        }
        launch_colmajor_to_inter(src, h->d, n, xi.p, n, st, h->perm());
        std::vector<cplx> tab((size_t)n * h->nplanes), hn(n);
        std::vector<zc> pc;
        auto pass = [&](int only_term, std::vector<double> &out) {     // norms of (sum_k c_jk A_k v_j), k = all or one term
            std::vector<double> ck((size_t)2 * T);
            for (int j = 0; j < n; ++j) {
                for (int k = 0; k < T; ++k) {
                    const bool on = only_term < 0 || k == only_term;
                    ck[2 * k] = on ? coeff_table[((size_t)j * T + k) * 2] : 0.0;
                    ck[2 * k + 1] = on ? coeff_table[((size_t)j * T + k) * 2 + 1] : 0.0;
                }
                plane_coeffs(h, ck.data(), WAE_OP_N, pc);
                for (int q = 0; q < h->nplanes; ++q) { const zc c = pc[h->slot_plane[0][q]]; tab[(size_t)j * h->nplanes + q] = cplx{c.real(), c.imag()}; }
            }
            pcd.upload(tab.data(), tab.size(), st);
            launch_spmv(h->ops[0].dev(WAE_OP_N), pcd.p, 1, xi.p, yi.p, nullptr, 0.0, n, MODE_AX, st);
            launch_norms(yi.p, h->d, n, part.p, nrm.p, st);
            HIP_CHECK(hipMemcpyAsync(hn.data(), nrm.p, (size_t)n * sizeof(cplx), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            out.resize(n);
            for (int j = 0; j < n; ++j) out[j] = hn[j].x;
        };
        std::vector<double> num, one, den(n, 0.0);
        pass(-1, num);
        for (int k = 0; k < T; ++k) {
            bool used = false;
            for (int j = 0; j < n && !used; ++j) used = coeff_table[((size_t)j * T + k) * 2] != 0.0 || coeff_table[((size_t)j * T + k) * 2 + 1] != 0.0;
            if (!used) continue;
            pass(k, one);
            for (int j = 0; j < n; ++j) den[j] += one[j];
        }
        for (int j = 0; j < n; ++j) res_out[j] = num[j] / std::max(den[j], 1e-300);
        xi.release(); yi.release(); pcd.release(); nrm.release(); part.release();
        return WAE_OK;
    });
}

int wae_spmv_sum(wae_family *h, const double *coeffs, const double *X, double *Y, int32_t r, int32_t op) {
    return wae_spmv_sum_cols(h, coeffs, 1, X, Y, r, op);
}

int wae_spmv_sum_multi(wae_family *h, const double *coeffs, const double *X, double *Y) {
    return guarded([&]() {
        WAE_REQUIRE(h && coeffs && X && Y, "bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        // a term aliased onto a shared plane still multiplies its own input column, so expand per TERM:
        // Y = sum_k c_k A_k X[:,k].  Terms sharing a plane are handled by one pass per distinct input column.
        const int T = h->T;
        const size_t cnt = (size_t)h->d * T;
        ensure(h->io_a, cnt); ensure(h->io_b, (size_t)h->d);
        DevBuf<cplx> xi, yi, acc, pcd;
        xi.alloc(cnt); yi.alloc((size_t)h->d); acc.alloc((size_t)h->d);
        HIP_CHECK(hipMemcpyAsync(h->io_a.p, X, cnt * sizeof(cplx), hipMemcpyHostToDevice, st));
        launch_colmajor_to_inter(h->io_a.p, h->d, T, xi.p, T, st, h->perm());
        launch_fill_zero(acc.p, (size_t)h->d, st);
        // passes: in pass t every plane takes the t-th term mapped to it (if any)
        std::vector<std::vector<int>> plane_terms(h->nplanes);
        for (int k = 0; k < T; ++k) plane_terms[h->term_plane[k]].push_back(k);
        size_t npass = 0;
        for (auto &v : plane_terms) npass = std::max(npass, v.size());
        std::vector<cplx> tab(h->nplanes);
        std::vector<int> pcol(h->nplanes);
        for (size_t t = 0; t < npass; ++t) {
            for (int s = 0; s < h->nplanes; ++s) {
                const int q = h->slot_plane[0][s];
                if (t < plane_terms[q].size()) {
                    const int k = plane_terms[q][t];
                    const zc c = h->term_scale[k] * zc(coeffs[2 * k], coeffs[2 * k + 1]);
                    tab[s] = cplx{c.real(), c.imag()};
                    pcol[s] = k;
                } else { tab[s] = cplx{0.0, 0.0}; pcol[s] = 0; }
            }
            pcd.upload(tab.data(), tab.size(), st);
            h->plane_col_dev.upload(pcol.data(), pcol.size(), st);
            launch_spmv_multi(h->ops[0].dev(WAE_OP_N), pcd.p, h->plane_col_dev.p, xi.p, yi.p, T, st);
            launch_add(yi.p, acc.p, (size_t)h->d, st);
            HIP_CHECK(hipStreamSynchronize(st));
        }
        launch_inter_to_colmajor(acc.p, 1, h->d, 1, h->io_b.p, st, h->perm());      // back to the caller's row numbering
        HIP_CHECK(hipMemcpyAsync(Y, h->io_b.p, (size_t)h->d * sizeof(cplx), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        xi.release(); yi.release(); acc.release(); pcd.release();
        return WAE_OK;
    });
}

int wae_solver_setup(wae_family *h, const double *coeffs_ref, const double *opts, int32_t nopts) {
    return guarded([&]() {
        WAE_REQUIRE(h && coeffs_ref, "bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        AmgOptions ao;
        auto opt = [&](int i, double dflt) { return (opts && i < nopts && opts[i] > 0) ? opts[i] : dflt; };
        ao.theta = opt(0, 0.02);
        ao.max_coarse = (int64_t)opt(1, 128);
        h->jac_w = opt(2, 0.8);
        // Post-smoothing and light-cycle weights (round 4; measured at 1M unknowns, pass in seconds, pre / post / light): 0.8 / 0.8 / 0.8
        // 2.02; 0.8 / 0.9 / 0.8 1.93; 0.9 / 1.0 / 0.8 2.07 (better snapshot solves, worse projected ones); 0.8 / 0.9 / 0.65 1.88;
        // 0.8 / 0.9 / 0.5 1.84; 0.8 / 0.9 / 0.3 1.80.  The light cycle's ONE sweep wants a small weight: its job is only to keep the
        // coarse correction honest on the components the coarse level cannot see.  0.5 is the default (eigenpair residuals and rank
        // gap of the benchmark unchanged: 6.6e-9, 1.3e9).
        h->jac_w_post = opt(10, 0.9);
        h->jac_w_light = opt(11, 0.5);
        h->nsweeps = (int)opt(3, 1);
        h->restart = (int)opt(4, 30);
        ao.penalty_ratio = opt(5, 1e8);
        h->NB = (int)opt(6, 64);
        WAE_REQUIRE(h->NB >= 1 && h->NB <= 256, "batch width must be in 1..256");
        WAE_REQUIRE(h->restart >= 2 && h->restart <= 200, "restart must be in 2..200");
        // drop a previous hierarchy
        for (size_t l = 1; l < h->ops.size(); ++l) {
            h->ops[l].diag.release();
            for (auto &G : h->ops[l].groups) { G.rowptr.release(); G.col.release(); G.rowptr_t.release(); G.col_t.release(); G.vals.release(); G.vals_t.release(); }
        }
        h->ops.resize(1);
        h->slot_plane.resize(1);
        for (auto &X : h->xfer) { X.p_ptr.release(); X.p_col.release(); X.r_ptr.release(); X.r_col.release(); X.p_val.release(); X.r_val.release(); }
        h->xfer.clear();
        std::vector<zc> pc;
        plane_coeffs(h, coeffs_ref, WAE_OP_N, pc);
        std::vector<AmgLevel> lv;
        std::vector<char> pen;
        // opts[7]: bit k set = term k stays out of the shape matrix (strength graph, aggregation, prolongator smoothing)
        const uint64_t excl = (uint64_t)opt(7, 0.0);
        std::vector<zc> pc_shape;
        if (excl) {
            std::vector<double> cs(coeffs_ref, coeffs_ref + (size_t)2 * h->T);
            for (int k = 0; k < h->T && k < 52; ++k)
                if (excl >> k & 1) cs[2 * k] = cs[2 * k + 1] = 0.0;
            plane_coeffs(h, cs.data(), WAE_OP_N, pc_shape);
        }
        const double t_amg0 = now_s();
        // The Krylov basis -- (restart + 1) vectors of d x NB complex numbers, 42 GB at 1M unknowns -- takes the driver about a
        // second to map: it is requested now, on a helper thread, and is there when the host part of the set-up is done.
        // opts[8], opts[9] (hints): probe columns and snapshot capacity of the contour integrals to come -- their snapshot store
        // (5 GB at 1M unknowns x 8 columns x 40 snapshots) and the resident term products (20 GB) are then mapped here as well,
        // behind the host work, instead of in the first pass (0.4 s of its snapshot phase).
        std::future<void> basis_job;
        {
            const size_t vec = (size_t)h->d * h->NB, need = vec * (size_t)(h->restart + 1);
            const size_t hint_l = (size_t)opt(8, 0.0), hint_s = (size_t)opt(9, 0.0);
            const size_t need_q = hint_l > 0 && hint_s > 0 && hint_l <= (size_t)h->NB ? (size_t)h->d * hint_l * hint_s : 0;
            const size_t need_w = need_q * (size_t)h->nplanes;
            h->rb.wait_w();
            if (h->V.n != need || h->rbQ.n < need_q || h->rb.W.n < need_w)
                basis_job = std::async(std::launch::async, [h, need, need_q, need_w]() {
                    HIP_CHECK(hipSetDevice(h->device));
                    if (h->V.n != need) h->V.alloc(need);
                    // (the snapshot stores are written once here as well: the first kernels that touch freshly mapped device memory
                    // ran slower -- 0.14 s over the first pass's snapshot phase on some boxes; behind the host work it costs nothing)
                    const bool new_q = h->rbQ.n < need_q, new_w = h->rb.W.n < need_w;
                    if (new_q) h->rbQ.alloc(need_q);
                    if (new_w) h->rb.W.alloc(need_w);
                    if (new_q && need_q) HIP_CHECK(hipMemset(h->rbQ.p, 0, need_q * sizeof(cplx)));
                    if (new_w && need_w) HIP_CHECK(hipMemset(h->rb.W.p, 0, need_w * sizeof(cplx)));
                    HIP_CHECK(hipDeviceSynchronize());
                });
        }
        struct Join { std::future<void> &f; ~Join() { if (f.valid()) f.wait(); } } basis_join{basis_job};     // (also on an exception)
        // fine-level aggregation in the caller's node order (iperm[o] = internal index of the caller's node o)
        std::vector<int> visit0;
        if (!h->perm_h.empty()) { visit0.resize(h->perm_h.size()); for (size_t i = 0; i < h->perm_h.size(); ++i) visit0[h->perm_h[i]] = (int)i; }
        // Everything of level 1 -- tile plan, renumbering, operator groups, tile storage, the transfer operators of level 0 and the tile
        // storage of the restriction -- needs only that level's planes and the first prolongator: it runs on a helper thread, with a
        // stream of its own, while the host builds the deeper levels (round 3, second half: 0.7 s of work of which 0.15 s used to be
        // hidden).  The level's operator is built into a local object and moved into the handle once the number of levels is known.
        const int tile1 = getenv("WAE_TILE_LEVEL1") ? atoi(getenv("WAE_TILE_LEVEL1")) : 1;      // (read per call: the tests switch it)
        const bool want_plan = tile1 && !h->tile_row_ptr.empty();
        struct Level1Work {
            std::vector<int> row_ptr, perm, iperm;      // tile plan of level 1 (empty: no tiles)
            int wmax = 0;
            std::vector<CsrZ> planes;                    // the level's planes in the new numbering (amg_setup still reads the old ones)
            LevelOp op;                                  // level-1 operator (groups + tiles)
            std::vector<int> slot_plane;
            Transfer xfer0;                              // P / R of level 0 (+ restriction tiles)
            bool built = false;
            double seconds = 0.0;
        } l1;
        std::future<void> l1_job;
        struct JoinL1 { std::future<void> &f; ~JoinL1() { if (f.valid()) f.wait(); } } l1_join{l1_job};
        auto upload_transfer = [](Transfer &X, const AmgLevel &L, hipStream_t s2) {
            X.nf = L.P.n; X.nc = L.P.m;
            X.p_ptr.upload(L.P.ptr.data(), L.P.ptr.size(), s2);
            X.p_col.upload(L.P.col.data(), L.P.col.size(), s2);
            X.p_val.upload(L.P.val.data(), L.P.val.size(), s2);
            X.r_ptr.upload(L.R.ptr.data(), L.R.ptr.size(), s2);
            X.r_col.upload(L.R.col.data(), L.R.col.size(), s2);
            X.r_val.upload(L.R.val.data(), L.R.val.size(), s2);
            HIP_CHECK(hipStreamSynchronize(s2));
        };
        auto rename_cols = [](CsrD &A, const std::vector<int> &ip) {            // column c -> ip[c], rows re-sorted
            std::vector<std::pair<int, double>> row;
            for (int64_t i = 0; i < A.n; ++i) {
                row.clear();
                for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) row.emplace_back(ip[A.col[p]], A.val[p]);
                std::sort(row.begin(), row.end(), [](const std::pair<int, double> &x, const std::pair<int, double> &y) { return x.first < y.first; });
                for (int p = A.ptr[i], k = 0; p < A.ptr[i + 1]; ++p, ++k) { A.col[p] = row[k].first; A.val[p] = row[k].second; }
            }
        };
        auto permute_rows = [](CsrD &A, const std::vector<int> &pm) {           // new row i = old row pm[i]
            CsrD B;
            B.n = A.n; B.m = A.m;
            B.ptr.assign(A.n + 1, 0);
            B.col.reserve(A.col.size()); B.val.reserve(A.val.size());
            for (int64_t i = 0; i < A.n; ++i) {
                const int o = pm[i];
                B.col.insert(B.col.end(), A.col.begin() + A.ptr[o], A.col.begin() + A.ptr[o + 1]);
                B.val.insert(B.val.end(), A.val.begin() + A.ptr[o], A.val.begin() + A.ptr[o + 1]);
                B.ptr[i + 1] = (int)B.col.size();
            }
            A = std::move(B);
        };
        amg_setup(h->planes0, pc, ao, lv, &pen, excl ? &pc_shape : nullptr, visit0.empty() ? nullptr : &visit0,
                  [&](const AmgLevel &L) {
                      if (!want_plan || l1_job.valid() || &L != &lv[0]) return;
                      // (the level object stays where it is -- amg_setup reserves its levels -- and nothing else touches it until the join)
                      l1_job = std::async(std::launch::async, [&, h]() {
                          const double tq0 = now_s();
                          HIP_CHECK(hipSetDevice(h->device));
                          hipStream_t s3;
                          HIP_CHECK(hipStreamCreate(&s3));
                          struct Del { hipStream_t s; ~Del() { (void)hipStreamDestroy(s); } } del{s3};
                          const int wcap = getenv("WAE_TILE_WCAP1") ? atoi(getenv("WAE_TILE_WCAP1")) : 608;        // (two window buffers)
                          const int thick = getenv("WAE_TILE_THICK") ? atoi(getenv("WAE_TILE_THICK")) : 6;
                          AmgLevel &L0 = lv[0];
                          const bool jdbg = getenv("WAE_SETUP_DEBUG") != nullptr;
                          double tj = now_s();
                          auto jlap = [&](const char *what) { if (jdbg) { const double t = now_s(); fprintf(stderr, "[setup]   (level-1 thread) %-22s %.3f s\n", what, t - tj); tj = t; } };
                          TilePlan plan = plan_tiles(union_pattern(L0.coarse_planes), 128, wcap, thick);
                          l1.wmax = plan.wmax;
                          jlap("tile plan");
                          if (!plan.perm.empty()) {
                              // Level 1 renumbered into tiles as well (the numbering of a coarse level is nobody's business but the
                              // hierarchy's): P of level 0 changes its columns, R its rows; the transfer to level 2 the other way round
                              // (after the join, when it exists).
                              l1.perm = plan.perm; l1.iperm = plan.iperm; l1.row_ptr = plan.row_ptr;
                              std::vector<std::future<void>> pj;
                              l1.planes.resize(L0.coarse_planes.size());
                              for (size_t q = 0; q < L0.coarse_planes.size(); ++q)
                                  pj.push_back(std::async(std::launch::async, [&L0, q, this_l1 = &l1]() {
                                      this_l1->planes[q] = permute_symmetric(L0.coarse_planes[q], this_l1->perm, this_l1->iperm);
                                  }));
                              auto j1 = std::async(std::launch::async, [&]() { rename_cols(L0.P, l1.iperm); });
                              permute_rows(L0.R, l1.perm);
                              j1.get();
                              for (auto &j : pj) j.get();
                          }
                          jlap("permutation");
                          // the transfer operators and the restriction's tile storage beside the operator (a thread and a stream of their own)
                          const bool with_tiles = !l1.row_ptr.empty();
                          auto xj = std::async(std::launch::async, [&, h, with_tiles, wcap]() {
                              HIP_CHECK(hipSetDevice(h->device));
                              hipStream_t s2;
                              HIP_CHECK(hipStreamCreate(&s2));
                              struct Del2 { hipStream_t s; ~Del2() { (void)hipStreamDestroy(s); } } del2{s2};
                              upload_transfer(l1.xfer0, L0, s2);
                              const int tile_r = getenv("WAE_TILE_RESTRICT") ? atoi(getenv("WAE_TILE_RESTRICT")) : 1;
                              if (with_tiles && tile_r) build_restriction_tiles(l1.xfer0, L0.R, wcap, s2);
                              if (!h->tile_row_ptr.empty() && xfer_tiles_on()) build_transfer_tiles(l1.xfer0, L0.P, h->tile_row_ptr, s2);
                          });
                          const std::vector<CsrZ> &pl1 = l1.planes.empty() ? L0.coarse_planes : l1.planes;
                          l1.slot_plane = build_levelop(l1.op, pl1, s3, WAE_LEVEL_SYM_TOL);
                          jlap("operator groups");
                          if (with_tiles) build_level_tiles(l1.op, pl1, l1.slot_plane, l1.row_ptr, s3, 4);
                          HIP_CHECK(hipStreamSynchronize(s3));
                          jlap("tile storage");
                          xj.get();
                          jlap("wait for the transfer");
                          l1.built = true;
                          l1.seconds = now_s() - tq0;
                      });
                  });
        const double t_amg1 = now_s();
        if (getenv("WAE_SETUP_DEBUG")) {
            fprintf(stderr, "[setup] amg_setup (host) %.3f s\n", t_amg1 - t_amg0);
            fprintf(stderr, "[setup] level 0: n=%lld nnz/plane:", (long long)h->planes0[0].n);
            for (const CsrZ &A : h->planes0) fprintf(stderr, " %lld", (long long)A.nnz());
            fprintf(stderr, "\n");
            for (size_t l = 0; l < lv.size(); ++l) {
                fprintf(stderr, "[setup] level %zu: n=%lld P nnz=%lld nnz/plane:", l + 1, (long long)lv[l].P.m, (long long)lv[l].P.col.size());
                for (const CsrZ &A : lv[l].coarse_planes) fprintf(stderr, " %lld", (long long)A.nnz());
                fprintf(stderr, "\n");
            }
        }
        hipStream_t st = h->stream;
        double t_lap = now_s();
        auto lap = [&](const char *what) {
            if (!getenv("WAE_SETUP_DEBUG")) return;
            HIP_CHECK(hipStreamSynchronize(st));
            const double t = now_s();
            fprintf(stderr, "[setup] %-34s %.3f s\n", what, t - t_lap);
            t_lap = t;
        };
        if (l1_job.valid()) l1_job.get();                                   // (rethrows)
        if (!l1.planes.empty()) lv[0].coarse_planes = std::move(l1.planes);
        if (l1.built && !l1.perm.empty() && lv.size() >= 2) {              // the transfer to level 2 in the new numbering of level 1
            auto j3 = std::async(std::launch::async, [&]() { permute_rows(lv[1].P, l1.perm); });
            rename_cols(lv[1].R, l1.iperm);
            j3.get();
        }
        const bool l1_dense = lv.size() < 2;                                // (a two-level hierarchy: level 1 is the dense one; its tiles are not used)
        if (getenv("WAE_SETUP_DEBUG") && l1.built)
            fprintf(stderr, "[setup] level 1 on the helper thread: plan + permutation + operator + tiles + transfer %.3f s (%zu tiles, largest window %d)\n",
                    l1.seconds, l1.row_ptr.empty() ? (size_t)0 : l1.row_ptr.size() - 1, l1.wmax);
        lap("wait for level 1 (helper thread)");
        {   // the penalty rows' own sub-block, plane by plane (compact numbering)
            std::vector<int> rows, loc(pen.size(), -1);
            for (size_t i = 0; i < pen.size(); ++i)
                if (pen[i]) { loc[i] = (int)rows.size(); rows.push_back((int)i); }
            h->n_penalty = (int64_t)rows.size();
            if (!rows.empty()) {
                std::vector<CsrZ> sub(h->planes0.size());
                for (size_t q = 0; q < h->planes0.size(); ++q) {
                    const CsrZ &A = h->planes0[q];
                    CsrZ &B = sub[q];
                    B.n = B.m = (int64_t)rows.size();
                    B.ptr.assign(rows.size() + 1, 0);
                    for (size_t i = 0; i < rows.size(); ++i) {
                        for (int pp = A.ptr[rows[i]]; pp < A.ptr[rows[i] + 1]; ++pp)
                            if (loc[A.col[pp]] >= 0) { B.col.push_back(loc[A.col[pp]]); B.val.push_back(A.val[pp]); }
                        B.ptr[i + 1] = (int)B.col.size();
                    }
                }
                h->pen_slot = build_levelop(h->pen_op, sub, st, WAE_LEVEL_SYM_TOL);
                for (size_t q = 0; q < h->planes0.size(); ++q) {      // the same rows with ALL their columns (global numbering)
                    const CsrZ &A = h->planes0[q];
                    CsrZ &B = sub[q];
                    B = CsrZ();
                    B.n = (int64_t)rows.size();
                    B.m = A.m;
                    B.ptr.assign(rows.size() + 1, 0);
                    for (size_t i = 0; i < rows.size(); ++i) {
                        B.col.insert(B.col.end(), A.col.begin() + A.ptr[rows[i]], A.col.begin() + A.ptr[rows[i] + 1]);
                        B.val.insert(B.val.end(), A.val.begin() + A.ptr[rows[i]], A.val.begin() + A.ptr[rows[i] + 1]);
                        B.ptr[i + 1] = (int)B.col.size();
                    }
                }
                h->pen_row_slot = build_levelop(h->pen_row_op, sub, st, WAE_LEVEL_SYM_TOL);
                h->pen_rows.upload(rows.data(), rows.size(), st);
                const size_t cnt = rows.size() * (size_t)h->NB;
                h->pen_b.alloc(cnt); h->pen_x.alloc(cnt); h->pen_t.alloc(cnt);
                HIP_CHECK(hipStreamSynchronize(st));
            }
        }
        lap("penalty operators");
        h->ops.resize(lv.size() + 1);
        h->slot_plane.resize(lv.size() + 1);
        h->xfer.resize(lv.size());
        for (size_t l = 0; l < lv.size(); ++l) {
            if (l == 0 && l1.built) {
                h->slot_plane[1] = l1.slot_plane;
                if (l1_dense) l1.op.tiles = TileStore();                   // (not reached in practice: a tiled fine level has a large level 1)
                h->ops[1] = std::move(l1.op);
                h->xfer[0] = std::move(l1.xfer0);
                continue;
            }
            h->slot_plane[l + 1] = build_levelop(h->ops[l + 1], lv[l].coarse_planes, st, WAE_LEVEL_SYM_TOL);
            upload_transfer(h->xfer[l], lv[l], st);
            if (l == 0 && !h->tile_row_ptr.empty() && xfer_tiles_on()) build_transfer_tiles(h->xfer[0], lv[0].P, h->tile_row_ptr, st);
        }
        lap("levels >= 2");
        // dense planes of the coarsest level (plane order, row-major)
        const std::vector<CsrZ> &last = lv.empty() ? h->planes0 : lv.back().coarse_planes;
        h->nc = last[0].n;
        WAE_REQUIRE(h->nc <= 2048, "coarsest level too large for the dense solver (increase levels / lower max_coarse)");
        {
            const size_t nn = (size_t)h->nc * h->nc;
            std::vector<cplx> dp(nn * h->nplanes, cplx{0.0, 0.0});
            const std::vector<int> &sp = h->slot_plane.back();
            for (int s = 0; s < h->nplanes; ++s) {
                const CsrZ &A = last[sp[s]];
                for (int64_t i = 0; i < A.n; ++i)
                    for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) dp[(size_t)s * nn + (size_t)i * h->nc + A.col[p]] = cplx{A.val[p].real(), A.val[p].imag()};
            }
            h->dense_planes.upload(dp.data(), dp.size(), st);
            HIP_CHECK(hipStreamSynchronize(st));
            h->Ainv.alloc(nn * h->NB);
            h->dstatus.alloc(1);
        }
        lap("dense coarsest level");
        // workspaces
        const int NB = h->NB, m = h->restart;
        const int nl = (int)h->ops.size();
        h->lx.resize(nl); h->lb.resize(nl); h->lt.resize(nl);
        for (int l = 0; l < nl; ++l) {
            const size_t cnt = (size_t)h->ops[l].n * NB;
            h->lx[l].alloc(cnt); h->lb[l].alloc(cnt); h->lt[l].alloc(cnt);
        }
        const size_t vec = (size_t)h->d * NB;
        lap("level workspaces");
        if (basis_job.valid()) basis_job.get();                      // (rethrows an allocation failure)
        lap("wait for the Krylov basis");
        if (h->V.n != vec * (m + 1)) h->V.alloc(vec * (m + 1));
        h->W.alloc(vec); h->Xs.alloc(vec); h->Bs.alloc(vec); h->U.alloc(vec);
        // masked (converged) columns keep stale data: make sure "stale" is never an uninitialised NaN pattern
        HIP_CHECK(hipMemsetAsync(h->V.p, 0, vec * (m + 1) * sizeof(cplx), st));
        HIP_CHECK(hipMemsetAsync(h->W.p, 0, vec * sizeof(cplx), st));
        HIP_CHECK(hipMemsetAsync(h->U.p, 0, vec * sizeof(cplx), st));
        for (int l = 0; l < nl; ++l) {
            const size_t cnt = (size_t)h->ops[l].n * NB;
            HIP_CHECK(hipMemsetAsync(h->lx[l].p, 0, cnt * sizeof(cplx), st));
            HIP_CHECK(hipMemsetAsync(h->lb[l].p, 0, cnt * sizeof(cplx), st));
            HIP_CHECK(hipMemsetAsync(h->lt[l].p, 0, cnt * sizeof(cplx), st));
        }
        HIP_CHECK(hipStreamSynchronize(st));
        h->partial.alloc((size_t)1024 * 32 * NB);   // DOT_BLOCKS x 32 vectors x NB columns
        h->hdev.alloc((size_t)2 * (m + 3) * NB);     // second half: scratch for the re-orthogonalisation pass
        h->vsq.alloc((size_t)(m + 3) * NB);
        h->ydev.alloc((size_t)(m + 1) * NB);
        if (h->h_pinned) { (void)hipHostFree(h->h_pinned); h->h_pinned = nullptr; }
        HIP_CHECK(hipHostMalloc((void **)&h->h_pinned, (size_t)(m + 2) * NB * sizeof(cplx)));
        h->solver_ready = true;
        lap("other workspaces + memsets");
        if (getenv("WAE_SETUP_DEBUG")) fprintf(stderr, "[setup] uploads + workspaces %.3f s\n", now_s() - t_amg1);
        return WAE_OK;
    });
}

// solve a chunk of nb columns already on device in interleaved layout
static int solve_chunk(wae_family *h, const Batch &bt, const std::vector<std::vector<zc>> &pcs, const cplx *B, cplx *X, double tol, int maxit,
                       wae_solve_info *info, const cplx *guess_dir = nullptr, bool have_x0 = false) {
    upload_pc(h, pcs);
    dense_setup(h, bt);
    return gmres(h, bt, B, X, tol, maxit, info, guess_dir, have_x0);
}

int wae_solve_guess(wae_family *h, const double *coeffs, int32_t ncoef, const double *B, const double *Gd, double *X, int32_t r, int32_t op,
                    double tol, int32_t maxit, wae_solve_info *info) {
    return guarded([&]() {
        WAE_REQUIRE(h && r >= 0 && (r == 0 || (coeffs && B && X)), "bad argument");
        WAE_REQUIRE(op >= 0 && op <= 2, "bad op");
        if (r == 0) {                                             // L(z) \ zeros(d, 0)
            if (info) std::memset(info, 0, sizeof(*info));
            return WAE_OK;
        }
        WAE_REQUIRE(ncoef == 1 || ncoef == r, "ncoef must be 1 or r");
        require_solver(h);
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        wae_solve_info li;
        memset(&li, 0, sizeof(li));
        const double t0 = now_s();
        const int64_t d = h->d;
        const size_t cnt = (size_t)d * r;
        ensure(h->io_a, cnt); ensure(h->io_b, cnt);
        HIP_CHECK(hipMemcpyAsync(h->io_a.p, B, cnt * sizeof(cplx), hipMemcpyHostToDevice, st));
        DevBuf<cplx> gcol, gint;
        if (Gd) {
            gcol.alloc(cnt);
            gint.alloc((size_t)d * h->NB);
            HIP_CHECK(hipMemcpyAsync(gcol.p, Gd, cnt * sizeof(cplx), hipMemcpyHostToDevice, st));
        }
        for (int c0 = 0; c0 < r; c0 += h->NB) {
            const int nb = std::min(h->NB, r - c0);
            Batch bt;
            bt.nb = nb; bt.op = op;
            std::vector<std::vector<zc>> pcs;
            if (ncoef == 1) {
                bt.cps = nb; bt.nsys = 1;
                pcs.resize(1);
                plane_coeffs(h, coeffs, op, pcs[0]);
            } else {
                bt.cps = 1; bt.nsys = nb;
                pcs.resize(nb);
                for (int b = 0; b < nb; ++b) plane_coeffs(h, coeffs + (size_t)(c0 + b) * 2 * h->T, op, pcs[b]);
            }
            launch_colmajor_to_inter(h->io_a.p + (size_t)c0 * d, d, nb, h->Bs.p, nb, st, h->perm());
            if (Gd) launch_colmajor_to_inter(gcol.p + (size_t)c0 * d, d, nb, gint.p, nb, st, h->perm());
            solve_chunk(h, bt, pcs, h->Bs.p, h->Xs.p, tol, maxit, &li, Gd ? gint.p : nullptr);
            launch_inter_to_colmajor(h->Xs.p, nb, d, nb, h->io_b.p + (size_t)c0 * d, st, h->perm());
        }
        HIP_CHECK(hipMemcpyAsync(X, h->io_b.p, cnt * sizeof(cplx), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        gcol.release(); gint.release();
        li.seconds = now_s() - t0;
        const int rc_ = info_code(li);
        if (info) *info = li;
        return rc_;
    });
}

int wae_solve(wae_family *h, const double *coeffs, int32_t ncoef, const double *B, double *X, int32_t r, int32_t op, double tol, int32_t maxit,
              wae_solve_info *info) {
    return wae_solve_guess(h, coeffs, ncoef, B, nullptr, X, r, op, tol, maxit, info);
}

int wae_beyn_moments(wae_family *h, int32_t npts, const double *z, const double *w, const double *coeff_table, const double *V, int32_t l, int32_t K,
                     double tol, int32_t maxit, double *A_out, uint64_t out_dev, wae_solve_info *info) {
    return guarded([&]() {
        WAE_REQUIRE(h && npts >= 0 && (npts == 0 || (z && w && coeff_table)) && V && l > 0 && K > 0, "bad argument");
        WAE_REQUIRE(A_out || out_dev, "no output buffer");
        require_solver(h);
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        wae_solve_info li;
        memset(&li, 0, sizeof(li));
        const double t0 = now_s();
        const int64_t d = h->d;
        const int npow = 2 * K;
        const size_t acnt = (size_t)d * l * npow;
        DevBuf<cplx> Aown;
        cplx *Ad = (cplx *)(uintptr_t)out_dev;
        if (!Ad) { Aown.alloc(acnt); Ad = Aown.p; }
        launch_fill_zero(Ad, acnt, st);
        ensure(h->io_a, (size_t)d * l);
        HIP_CHECK(hipMemcpyAsync(h->io_a.p, V, (size_t)d * l * sizeof(cplx), hipMemcpyHostToDevice, st));
        // the reference accepts any l (beyn.jl:39-57): probe columns beyond the batch width are handled in groups of <= NB
        for (int cg = 0; cg < l; cg += h->NB) {
            const int lg = std::min(h->NB, l - cg);
            const int spc = std::max(1, h->NB / lg);   // systems per chunk
            ensure(h->zw_dev, (size_t)2 * spc);
            int rep_nb = -1;
            for (int p0 = 0; p0 < npts; p0 += spc) {
                const int ns = std::min(spc, npts - p0);
                Batch bt;
                bt.nb = ns * lg; bt.cps = lg; bt.nsys = ns; bt.op = WAE_OP_N;
                std::vector<std::vector<zc>> pcs(ns);
                std::vector<cplx> zw(2 * ns);
                for (int s = 0; s < ns; ++s) {
                    plane_coeffs(h, coeff_table + (size_t)(p0 + s) * 2 * h->T, WAE_OP_N, pcs[s]);
                    zw[s] = cplx{w[2 * (p0 + s)], w[2 * (p0 + s) + 1]};
                    zw[ns + s] = cplx{z[2 * (p0 + s)], z[2 * (p0 + s) + 1]};
                }
                h->zw_dev.upload(zw.data(), zw.size(), st);
                HIP_CHECK(hipStreamSynchronize(st));
                if (bt.nb != rep_nb) { launch_replicate(h->io_a.p + (size_t)cg * d, d, lg, h->Bs.p, bt.nb, st, h->perm()); rep_nb = bt.nb; }   // same right-hand sides for every chunk
                solve_chunk(h, bt, pcs, h->Bs.p, h->Xs.p, tol, maxit, &li);
                launch_beyn_accum(h->Xs.p, bt.nb, d, lg, ns, h->zw_dev.p, h->zw_dev.p + ns, npow, Ad, st, l, cg, h->perm());
            }
        }
        if (A_out) HIP_CHECK(hipMemcpyAsync(A_out, Ad, acnt * sizeof(cplx), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        Aown.release();
        li.seconds = now_s() - t0;
        const int rc_ = info_code(li);
        if (info) *info = li;
        return rc_;
    });
}

int wae_beyn_moments_rb(wae_family *h, int32_t npts, const double *z, const double *w, const double *coeff_table, const double *V, int32_t l, int32_t K,
                        double tol, int32_t maxit, int32_t mode, int32_t nbasis, int32_t slot0, uint64_t Q_dev, double *A_out, uint64_t out_dev,
                        int32_t accumulate, int32_t l_total, int32_t col0, wae_solve_info *info) {
    return guarded([&]() {
        WAE_REQUIRE(h && npts >= 0 && (npts == 0 || (z && w && coeff_table)) && (V || mode == 2) && l > 0 && K > 0, "bad argument");
        WAE_REQUIRE(A_out || out_dev, "no output buffer");
        WAE_REQUIRE(mode >= 0 && mode <= 4, "mode must be 0 (take snapshots), 1 (rebuild the basis from the store, use it), 2 (use it), "
                                            "3 (solve from zero, store raw) or 4 (build the basis from the store, solve nothing)");
        WAE_REQUIRE(nbasis >= 0 && slot0 >= 0 && (mode == 2 || slot0 + ((mode == 0 || mode == 3) ? npts : 0) <= nbasis), "snapshot slots out of range");
        WAE_REQUIRE(!accumulate || out_dev, "accumulate needs a device-resident moment buffer");
        if (l_total <= 0) { l_total = l; col0 = 0; }
        WAE_REQUIRE(col0 >= 0 && col0 + l <= l_total, "column slice out of range");
        require_solver(h);
        WAE_REQUIRE(l <= h->NB, "l exceeds the solver batch width");
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        wae_solve_info li;
        memset(&li, 0, sizeof(li));
        const double t0 = now_s();
        const int64_t d = h->d;
        const int npow = 2 * K;
        const size_t acnt = (size_t)d * l_total * npow;      // the moment tensor has l_total columns; this call fills l of them
        const size_t vecl = (size_t)d * l;
        const int T = h->T;
        RbState &R = h->rb;
        DevBuf<cplx> Aown;
        cplx *Ad = (cplx *)(uintptr_t)out_dev;
        if (!Ad) { Aown.alloc(acnt); Ad = Aown.p; }
        if (!accumulate) launch_fill_zero(Ad, acnt, st);
        cplx *Q = (cplx *)(uintptr_t)Q_dev;
        if (!Q) {                                    // library-owned snapshot store (single-process use)
            if (mode == 0 && slot0 == 0 && h->rbQ.n < vecl * (size_t)nbasis) h->rbQ.alloc(vecl * (size_t)nbasis);
            WAE_REQUIRE(h->rbQ.n >= vecl * (size_t)nbasis, "no snapshots stored in the handle: run mode 0 first");
            Q = h->rbQ.p;
        }
        ensure(h->io_a, vecl);
        if (V) {
            HIP_CHECK(hipMemcpyAsync(h->io_a.p, V, vecl * sizeof(cplx), hipMemcpyHostToDevice, st));
        } else {                                     // mode 2 on the basis this handle started: its probe matrix is still in HBM
            WAE_REQUIRE(mode == 2 && R.vi_valid && R.l == l && R.Vi.n >= vecl, "V may be NULL only in mode 2 after a mode 0/1 call with the same l on this handle");
            launch_inter_to_colmajor(R.Vi.p, l, d, l, h->io_a.p, st, h->perm());      // (io_a holds the caller's numbering, like an uploaded V)
        }
        const int spc = std::max(1, h->NB / l);   // systems per chunk
        ensure(h->zw_dev, (size_t)2 * spc);
        if (R.ycoef.n < (size_t)std::max(nbasis, 1) * h->NB) R.ycoef.alloc((size_t)std::max(nbasis, 1) * h->NB);

        static const bool rbdbg = getenv("WAE_GMRES_DEBUG") && atoi(getenv("WAE_GMRES_DEBUG"));
        // adaptive enrichment is off by default: on the C2 contour the orthogonalisation and projection of the extra
        // vectors cost more than the iterations they saved (measured with thresholds 3, 6, 9); WAE_RB_ENRICH=<its> enables
        static const int enrich_its = getenv("WAE_RB_ENRICH") ? atoi(getenv("WAE_RB_ENRICH")) : (1 << 30);
        double t_guess = 0.0, t_solve = 0.0, t_append = 0.0;
        if ((mode == 0 && slot0 == 0) || mode == 1 || mode == 4) {
            launch_colmajor_to_inter(h->io_a.p, d, l, h->W.p, l, st, h->perm());
            rb_reset(h, Q, nbasis, l, coeff_table, npts, h->W.p);
            if (mode == 1 || mode == 4) rb_append(h, slot0);    // the store holds slot0 raw snapshots (e.g. gathered from other ranks)
            if (mode == 4) {                       // basis built (the coefficient table only said which terms take part): nothing to solve
                HIP_CHECK(hipStreamSynchronize(st));
                Aown.release();
                li.seconds = now_s() - t0;
                li.levels = (int)h->ops.size();
                if (info) *info = li;
                return (int)WAE_OK;
            }
        } else if (mode == 3) {
            // raw snapshots: the handle's basis is not touched
        } else {
            WAE_REQUIRE(R.Q == Q && R.l == l && R.cap == nbasis, "the basis in the handle belongs to another store / shape");
            WAE_REQUIRE(mode != 0 || slot0 == R.S, "mode 0 appends: slot0 must equal the number of snapshots taken so far");
            WAE_REQUIRE(mode != 2 || R.S > 0, "mode 2 needs a basis: run mode 0 (or 1) first");
        }

        // with a fixed basis (modes 1/2, no enrichment) the coefficients of the next chunk's guesses are computed on a helper
        // thread while the device solves the current chunk
        const bool fixed_basis = (mode == 1 || mode == 2) && enrich_its >= (1 << 30);
        std::future<std::vector<cplx>> next_Y;
        auto launch_coeffs = [&](int q0) {
            const int nq = std::min(spc, npts - q0);
            return std::async(std::launch::async, [h, coeff_table, q0, nq, T]() { return rb_guess_coeffs(h, coeff_table + (size_t)q0 * 2 * T, nq); });
        };
        if (fixed_basis && npts > 0 && R.S > 0) next_Y = launch_coeffs(0);
        // Mode 0 is progressive, so the first chunks should be small: a chunk never takes more points than the basis already
        // holds, starting with 16 columns' worth (1, 1, 2, 4, 4, ... points for l = 16; 16, 16, 32 for l = 1).  Measured on
        // the snapshot phase: C2 0.695 -> 0.668 s, C3 2.71 -> 2.56 s; one rank's share of C3 when the probe columns are split
        // over 8 / 4 / 2 GPUs (1 / 2 / 4 columns x 64 points): 0.97 -> 0.63, 1.19 -> 0.97, 1.79 -> 1.69 s (dev/c3_rank_share.py).
        // WAE_RB_DOUBLING=0 restores full chunks, WAE_RB_C0COLS sets the starting width (4, 8, 32 measured: slower).
        static const int doubling = getenv("WAE_RB_DOUBLING") ? atoi(getenv("WAE_RB_DOUBLING")) : 1;
        static const int c0cols = getenv("WAE_RB_C0COLS") ? std::max(1, atoi(getenv("WAE_RB_C0COLS"))) : 16;
        const int c0 = doubling ? std::max(1, c0cols / l) : spc;
        int ns_next = 0, rep_nb = -1;
        for (int p0 = 0; p0 < npts; p0 += ns_next) {
            const int ns = (mode == 0) ? std::min(std::min(spc, npts - p0), std::max(c0, R.S)) : std::min(spc, npts - p0);
            ns_next = ns;
            Batch bt;
            bt.nb = ns * l; bt.cps = l; bt.nsys = ns; bt.op = WAE_OP_N;
            std::vector<std::vector<zc>> pcs(ns);
            std::vector<cplx> zw(2 * ns);
            for (int s = 0; s < ns; ++s) {
                plane_coeffs(h, coeff_table + (size_t)(p0 + s) * 2 * T, WAE_OP_N, pcs[s]);
                zw[s] = cplx{w[2 * (p0 + s)], w[2 * (p0 + s) + 1]};
                zw[ns + s] = cplx{z[2 * (p0 + s)], z[2 * (p0 + s) + 1]};
            }
            h->zw_dev.upload(zw.data(), zw.size(), st);
            HIP_CHECK(hipStreamSynchronize(st));
            if (bt.nb != rep_nb) { launch_replicate(h->io_a.p, d, l, h->Bs.p, bt.nb, st, h->perm()); rep_nb = bt.nb; }   // same right-hand sides for every chunk
            const bool guess = mode != 3 && R.S > 0;   // mode 0 is progressive: later snapshot chunks start from the earlier ones
            const double ta = now_s();
            if (guess) {
                std::vector<cplx> Y;
                if (next_Y.valid()) {
                    Y = next_Y.get();
                    if (p0 + spc < npts) next_Y = launch_coeffs(p0 + spc);
                } else {
                    Y = rb_guess_coeffs(h, coeff_table + (size_t)p0 * 2 * T, ns);
                }
                rb_apply_guess(h, Y, ns, h->Xs.p);
            }
            const double tb = now_s();
            struct LightGuard { wae_family *h; ~LightGuard() { h->vc_light = false; } } light_guard{h};
            h->vc_light = guess && (mode == 1 || mode == 2);       // (vcycle: the cheaper cycle for the solves of the projected phase)
            const int its = solve_chunk(h, bt, pcs, h->Bs.p, h->Xs.p, tol, maxit, &li, nullptr, guess);
            h->vc_light = false;
            launch_beyn_accum(h->Xs.p, bt.nb, d, l, ns, h->zw_dev.p, h->zw_dev.p + ns, npow, Ad, st, l_total, col0, h->perm());
            if (rbdbg) HIP_CHECK(hipStreamSynchronize(st));
            const double tc = now_s();
            // mode 0 keeps every solution; modes 1/2 enrich the basis where the guesses were poor (a region of the
            // contour close to poles outside it), as long as the store has room
            if (mode == 3) {                       // raw solutions into the caller's slots; no basis work (another rank builds it)
                for (int s = 0; s < ns; ++s)
                    launch_extract_cols(h->Xs.p, bt.nb, s * l, l, Q + (size_t)(slot0 + p0 + s) * vecl, d, st);
            } else
            if (mode == 0 || (its > enrich_its && R.S + ns <= R.cap)) {
                for (int s = 0; s < ns; ++s)
                    launch_extract_cols(h->Xs.p, bt.nb, s * l, l, Q + (size_t)(R.S + s) * vecl, d, st);
                rb_append(h, ns);
            }
            if (rbdbg) {
                HIP_CHECK(hipStreamSynchronize(st));
                t_guess += tb - ta; t_solve += tc - tb; t_append += now_s() - tc;
            }
        }
        if (rbdbg) fprintf(stderr, "[rb] mode=%d S=%d guess %.3f s  solve %.3f s  append %.3f s\n", mode, R.S, t_guess, t_solve, t_append);
        if (A_out) HIP_CHECK(hipMemcpyAsync(A_out, Ad, acnt * sizeof(cplx), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        Aown.release();
        li.seconds = now_s() - t0;
        const int rc_ = info_code(li);
        if (info) *info = li;
        return rc_;
    });
}

int wae_rb_export(wae_family *h, int32_t *S_out, int32_t *l_out, int32_t *nk_out, int32_t *kact_out, double *Hk_out, double *g_out) {
    return guarded([&]() {
        WAE_REQUIRE(h && S_out && l_out && nk_out, "bad argument");
        const RbState &R = h->rb;
        *S_out = R.S; *l_out = R.l; *nk_out = (int32_t)R.kact.size();
        if (kact_out) for (size_t i = 0; i < R.kact.size(); ++i) kact_out[i] = R.kact[i];
        const int S = R.S, l = R.l, cap = R.cap;
        if (Hk_out)                                              // dense [ki][s][i][c], c fastest
            for (size_t ki = 0; ki < R.kact.size(); ++ki)
                for (int s = 0; s < S; ++s)
                    for (int i = 0; i < S; ++i)
                        for (int c = 0; c < l; ++c) {
                            const zc v = R.Hk[ki * (size_t)cap * cap * l + ((size_t)s * cap + i) * l + c];
                            const size_t o = (((ki * S + s) * (size_t)S + i) * l + c) * 2;
                            Hk_out[o] = v.real(); Hk_out[o + 1] = v.imag();
                        }
        if (g_out)
            for (int i = 0; i < S; ++i)
                for (int c = 0; c < l; ++c) { g_out[((size_t)i * l + c) * 2] = R.g[(size_t)i * l + c].real(); g_out[((size_t)i * l + c) * 2 + 1] = R.g[(size_t)i * l + c].imag(); }
        return WAE_OK;
    });
}

int wae_rb_import(wae_family *h, int32_t S, int32_t l, uint64_t Q_dev, int32_t nk, const int32_t *kact, const double *Hk, const double *g) {
    return guarded([&]() {
        WAE_REQUIRE(h && S > 0 && l > 0 && Q_dev && nk >= 0 && (nk == 0 || kact) && Hk && g, "bad argument");
        RbState &R = h->rb;
        R.Q = (cplx *)(uintptr_t)Q_dev; R.cap = S; R.l = l; R.S = S;
        R.vi_valid = false;
        R.kact.assign(kact, kact + nk);
        R.Hk.resize((size_t)nk * S * S * l);
        for (size_t i = 0; i < R.Hk.size(); ++i) R.Hk[i] = zc(Hk[2 * i], Hk[2 * i + 1]);     // cap == S: same dense layout
        R.g.resize((size_t)S * l);
        for (size_t i = 0; i < R.g.size(); ++i) R.g[i] = zc(g[2 * i], g[2 * i + 1]);
        return WAE_OK;
    });
}

// relative Ritz residual |h_{k+1,k}| |y_k| / |theta| of the dominant Ritz pair of the leading k x k block of H (column-major,
// leading dimension ld), by power iteration on the small matrix; +inf when the dominant eigenvalue is not well separated
static double dominant_ritz_residual(const std::vector<zc> &H, int ld, int k) {
    std::vector<zc> y(k, zc(0)), w(k);
    y[0] = 1.0;
    zc theta = 0, prev = 0;
    for (int it = 0; it < 400; ++it) {
        for (int i = 0; i < k; ++i) {
            zc acc = 0;
            for (int j = std::max(0, i - 1); j < k; ++j) acc += H[(size_t)j * ld + i] * y[j];      // Hessenberg: H[i][j] = 0 for i > j+1
            w[i] = acc;
        }
        double nrm = 0.0;
        for (const zc &v : w) nrm += std::norm(v);
        nrm = std::sqrt(nrm);
        if (!(nrm > 0.0)) return INFINITY;
        theta = 0;
        for (int i = 0; i < k; ++i) theta += std::conj(y[i]) * w[i];
        for (int i = 0; i < k; ++i) y[i] = w[i] / nrm;
        if (it > 2 && std::abs(theta - prev) <= 1e-14 * std::abs(theta)) {
            return std::abs(H[(size_t)(k - 1) * ld + k]) * std::abs(y[k - 1]) / std::abs(theta);
        }
        prev = theta;
    }
    return INFINITY;
}

static const cplx *slot_cols_ptr(wae_family *h, int32_t slot, const int32_t *cols, int n, DevBuf<cplx> &stage, hipStream_t st);   // (slots: below)

// The Arnoldi processes behind wae_arnoldi_shiftinvert_batch (start vectors and basis through host memory) and
// wae_arnoldi_shiftinvert_slots (start vectors from a device-resident multivector, basis kept on the device for wae_arnoldi_ritz_to_slot).
// v0_host: d x nsys column-major in the caller's row numbering, or null: then v0_slot / v0_cols name the columns.  V_out null: the
// basis stays in h->arn_EV (h->arn_nsys systems, h->arn_cols vectors each).
static int arnoldi_core(wae_family *h, int32_t nsys, const double *coeffsA, const double *coeffsM, int32_t m, const double *v0_host, int32_t v0_slot,
                        const int32_t *v0_cols, int32_t op, double tol, int32_t maxit, double ritz_tol, double *H_out, double *V_out,
                        wae_solve_info *info) {
    {
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        wae_solve_info li;
        memset(&li, 0, sizeof(li));
        const double t0 = now_s();
        const int64_t d = h->d;
        const int T = h->T;
        const size_t vec = (size_t)d * nsys;
        // (work space of the family, grown on demand and kept: a hipMalloc / hipFree pair of 0.9 GB per call at 1M DoF and 8 systems otherwise)
        DevBuf<cplx> &EV = h->arn_EV, &t = h->arn_t, &pcM = h->arn_pcM, &hcol = h->arn_hcol, &stage = h->arn_stage, &gdir = h->arn_gdir;
        h->arn_nsys = 0; h->arn_cols = 0;
        ensure(EV, vec * (m + 1));
        ensure(t, vec);
        if (m > 1) ensure(gdir, vec);
        ensure(stage, vec);
        ensure(hcol, (size_t)2 * (m + 2) * nsys);
        // per-system plane coefficients of M (level-0 slot order) and of A (all levels)
        std::vector<cplx> tab((size_t)nsys * h->nplanes);
        std::vector<zc> pcm;
        std::vector<std::vector<zc>> pcs(nsys);
        for (int sy = 0; sy < nsys; ++sy) {
            plane_coeffs(h, coeffsM + (size_t)sy * 2 * T, op, pcm);
            for (int q = 0; q < h->nplanes; ++q) { const zc c = pcm[h->slot_plane[0][q]]; tab[(size_t)sy * h->nplanes + q] = cplx{c.real(), c.imag()}; }
            plane_coeffs(h, coeffsA + (size_t)sy * 2 * T, op, pcs[sy]);
        }
        pcM.upload(tab.data(), tab.size(), st);
        Batch bt;
        bt.nb = nsys; bt.cps = 1; bt.nsys = nsys; bt.op = op;
        upload_pc(h, pcs);
        dense_setup(h, bt);
        std::vector<std::vector<zc>> H(nsys, std::vector<zc>((size_t)(m + 1) * m, zc(0)));
        std::vector<char> dead(nsys, 0);
        // v_0 = v0 / ||v0||, column by column
        if (v0_host) {
            HIP_CHECK(hipMemcpyAsync(stage.p, v0_host, vec * sizeof(cplx), hipMemcpyHostToDevice, st));
            launch_colmajor_to_inter(stage.p, d, nsys, t.p, nsys, st, h->perm());
        } else {                                             // (slot columns are in the library's row numbering already)
            const cplx *S = slot_cols_ptr(h, v0_slot, v0_cols, nsys, stage, st);
            launch_colmajor_to_inter(S, d, nsys, t.p, nsys, st, nullptr);
        }
        launch_norms(t.p, d, nsys, h->partial.p, hcol.p, st);
        launch_scale_inv(t.p, hcol.p, EV.p, d, nsys, st);
        const OpDev Mop = h->ops[0].dev(op);
        std::vector<cplx> hh((size_t)(m + 2) * nsys), al(nsys);
        int done = 0;
        // Relaxed inner tolerance (inexact Arnoldi: Bouras & Fraysse 2005, Simoncini 2005): the solve of step k may be as
        // inexact as tol / (relative Ritz residual after step k-1) without the true residual of the Ritz pair leaving the
        // computed one by more than ~m tol -- the k-th column of H enters the wanted Ritz vector with a weight of that size.
        // Close to an eigenvalue of the NLEVP (residuals 1e-5, 1e-10 after one and two steps) the second and third solves
        // stop at 1e-8 and 1e-3 instead of 1e-12.  Only with the Ritz test on (ritz_tol > 0, where the residuals are
        // evaluated anyway); a tenth of the bound, never looser than 1e-3, never tighter than tol.  WAE_ARNOLDI_RELAX=0: off.
        static const bool relax_on = !(getenv("WAE_ARNOLDI_RELAX") && atoi(getenv("WAE_ARNOLDI_RELAX")) == 0);
        double tol_j = tol;
        // A poor start costs a whole process step at full accuracy: the left process of a Newton step on a non-symmetric family starts from
        // a vector that is not close to the left eigenvector (relative Ritz residual 0.99 after its first step at 1M DoF), that step's
        // solve ran 86 iterations to an estimate of 1e-12 whose recomputed residual stood at 2.5e-3 -- the deflation direction, the start
        // itself, was useless -- and all it gave was a better direction.  The quality of a start is known beforehand:
        // q = ||M^-1 A v0|| / ||v0|| (one product, one V-cycle; M^-1 A is close to the identity away from the operator's near-null space,
        // so q ~ 1e-6 for an eigenvector estimate of that accuracy and ~ 1 for an arbitrary vector).  Columns with q > 0.1 are replaced by
        // one step of inverse iteration from them, solved to 1e-3 only: the process then starts from a vector whose own step is worth its
        // accuracy.  Only with the Ritz test on (the Newton-type solvers); WAE_ARNOLDI_PRESTEP=0: off.
        static const bool prestep_on = !(getenv("WAE_ARNOLDI_PRESTEP") && atoi(getenv("WAE_ARNOLDI_PRESTEP")) == 0);
        if (prestep_on && ritz_tol > 0.0 && m > 1 && h->ops.size() > 1 && tol < 1e-3) {
            launch_spmv(h->ops[0].dev(op), pc_level(h, 0), bt.cps, EV.p, h->W.p, nullptr, 0.0, nsys, MODE_AX, st);
            launch_norms(vcycle(h, bt, 0, h->W.p), d, nsys, h->partial.p, hcol.p, st);
            HIP_CHECK(hipMemcpyAsync(hh.data(), hcol.p, (size_t)nsys * sizeof(cplx), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            std::vector<cplx> keep(nsys), take(nsys);
            int npoor = 0;
            for (int sy = 0; sy < nsys; ++sy) {
                const bool poor = hh[sy].x > 0.1;
                npoor += poor;
                keep[sy] = cplx{poor ? 0.0 : 1.0, 0.0};
                take[sy] = cplx{poor ? 1.0 : 0.0, 0.0};
            }
            if (getenv("WAE_GMRES_DEBUG")) {
                double qmax = 0.0;
                for (int sy = 0; sy < nsys; ++sy) qmax = std::max(qmax, hh[sy].x);
                fprintf(stderr, "[arnoldi] start quality max %.2e: %d of %d columns take a step of inverse iteration first\n", qmax, npoor, nsys);
                if (atoi(getenv("WAE_GMRES_DEBUG")) > 2) { fprintf(stderr, "[arnoldi]   q ="); for (int sy = 0; sy < nsys; ++sy) fprintf(stderr, " %.2e", hh[sy].x); fprintf(stderr, "\n"); }
            }
            if (npoor) {
                wae_solve_info lpre;
                memset(&lpre, 0, sizeof(lpre));
                launch_spmv(Mop, pcM.p, 1, EV.p, t.p, nullptr, 0.0, nsys, MODE_AX, st);
                // (no deflation direction: the start is the only candidate and deflating a vector that is NOT near the null space costs
                // iterations -- 93 against 50 at 1M DoF; the restarted wide recurrence stalls on these systems: 161 steps)
                gmres(h, bt, t.p, h->Xs.p, 1e-3, maxit, &lpre, nullptr);
                li.iters_max = std::max(li.iters_max, lpre.iters_max);
                li.iters_total += lpre.iters_total;
                launch_norms(h->Xs.p, d, nsys, h->partial.p, hcol.p, st);
                launch_scale_inv(h->Xs.p, hcol.p, t.p, d, nsys, st);                 // t = the normalised iterates
                h->ydev.upload(keep.data(), nsys, st);
                launch_mask_cols(EV.p, h->ydev.p, d, nsys, st);
                HIP_CHECK(hipStreamSynchronize(st));                                 // (ydev is read by the kernel: before the next upload)
                h->ydev.upload(take.data(), nsys, st);
                launch_mask_cols(t.p, h->ydev.p, d, nsys, st);
                launch_add(t.p, EV.p, vec, st);
                HIP_CHECK(hipStreamSynchronize(st));
            }
        }
        for (int j = 0; j < m; ++j) {
            launch_spmv(Mop, pcM.p, 1, EV.p + (size_t)j * vec, t.p, nullptr, 0.0, nsys, MODE_AX, st);
            // the start vector is the caller's estimate of the wanted eigenvector: deflated out of the first solve of the process; the
            // later solves deflate the (normalised) solution of the first one -- one step of inverse iteration closer to that
            // eigenvector, which matters when the start is poor (the left process of a Newton step starts from conj(v): for a spinning
            // mode of an annulus that is the OTHER mode of the pair)
            gmres(h, bt, t.p, h->Xs.p, tol_j, maxit, &li, j == 0 ? EV.p : gdir.p);
            cplx *w = h->Xs.p;
            if (j == 0 && m > 1) {
                launch_norms(w, d, nsys, h->partial.p, hcol.p, st);
                launch_scale_inv(w, hcol.p, gdir.p, d, nsys, st);
            }
            std::vector<std::vector<zc>> hc(nsys, std::vector<zc>(j + 2, zc(0)));
            for (int pass = 0; pass < 2; ++pass) {               // classical Gram-Schmidt, two passes, per column
                launch_dots(EV.p, vec, j + 1, w, d, nsys, h->partial.p, hcol.p, st);
                launch_axpy_neg(EV.p, vec, j + 1, hcol.p, w, d, nsys, st);
                HIP_CHECK(hipMemcpyAsync(hh.data(), hcol.p, (size_t)(j + 1) * nsys * sizeof(cplx), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                for (int sy = 0; sy < nsys; ++sy)
                    for (int i = 0; i <= j; ++i) hc[sy][i] += zc(hh[(size_t)i * nsys + sy].x, hh[(size_t)i * nsys + sy].y);
            }
            launch_norms(w, d, nsys, h->partial.p, hcol.p, st);
            HIP_CHECK(hipMemcpyAsync(hh.data(), hcol.p, (size_t)nsys * sizeof(cplx), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            done = j + 1;
            bool any_alive = false;
            for (int sy = 0; sy < nsys; ++sy) {
                double scale = 0.0;
                for (int i = 0; i <= j; ++i) scale = std::max(scale, std::abs(hc[sy][i]));
                const bool brk = dead[sy] || !(hh[sy].x > 1e-14 * scale);    // invariant subspace: this column stops here
                hc[sy][j + 1] = brk ? zc(0) : zc(hh[sy].x);
                if (!dead[sy]) for (int i = 0; i <= j + 1; ++i) H[sy][(size_t)j * (m + 1) + i] = hc[sy][i];
                if (brk) dead[sy] = 1;
                al[sy] = brk ? cplx{0.0, 0.0} : cplx{hh[sy].x, 0.0};
                any_alive = any_alive || !brk;
            }
            if (!any_alive) break;
            h->ydev.upload(al.data(), nsys, st);
            launch_scale_inv(w, h->ydev.p, EV.p + (size_t)(j + 1) * vec, d, nsys, st);
            HIP_CHECK(hipStreamSynchronize(st));
            // early exit: the dominant Ritz pair of every live process has converged (close to an eigenvalue of the NLEVP
            // two or three steps do; a fixed m = 6 spent twice the solves)
            if (ritz_tol > 0.0 && j + 1 < m) {
                double worst = 0.0;
                for (int sy = 0; sy < nsys; ++sy)
                    if (!dead[sy]) worst = std::max(worst, dominant_ritz_residual(H[sy], m + 1, j + 1));
                if (getenv("WAE_GMRES_DEBUG")) fprintf(stderr, "[arnoldi] step %d worst relative Ritz residual %.2e\n", j + 1, worst);
                if (worst <= ritz_tol) break;
                if (relax_on) tol_j = std::max(tol, std::min(1e-3, 0.1 * tol / worst));          // (worst = inf: tol)
            }
        }
        // V_out[sys] = d x (m+1) column-major; only the columns the processes produced are written (steps taken + 1): the rest
        // of the caller's buffer is left as it was (the H columns beyond them are zero) -- at 1M DoF and 8 systems every column is
        // 128 MB of host memory to touch
        if (V_out) {
            for (int j = 0; j <= std::min(done, m); ++j) {
                launch_inter_to_colmajor(EV.p + (size_t)j * vec, nsys, d, nsys, stage.p, st, h->perm());
                for (int sy = 0; sy < nsys; ++sy)
                    HIP_CHECK(hipMemcpyAsync(V_out + ((size_t)sy * (m + 1) + j) * d * 2, stage.p + (size_t)sy * d, (size_t)d * sizeof(cplx),
                                             hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
            }
            if (done < m && !(ritz_tol > 0.0))               // every process ended in an invariant subspace: the documented zeros
                for (int sy = 0; sy < nsys; ++sy)
                    memset(V_out + ((size_t)sy * (m + 1) + done + 1) * d * 2, 0, (size_t)(m - done) * d * sizeof(cplx));
        } else {
            h->arn_nsys = nsys;
            h->arn_cols = std::min(done, m) + 1;
        }
        for (int sy = 0; sy < nsys; ++sy) memcpy(H_out + (size_t)sy * (m + 1) * m * 2, H[sy].data(), H[sy].size() * sizeof(zc));
        li.seconds = now_s() - t0;
        const int rc_ = info_code(li);
        if (info) *info = li;
        return rc_;
    }
}

int wae_arnoldi_shiftinvert_batch(wae_family *h, int32_t nsys, const double *coeffsA, const double *coeffsM, int32_t m, const double *v0, int32_t op,
                                  double tol, int32_t maxit, double ritz_tol, double *H_out, double *V_out, wae_solve_info *info) {
    return guarded([&]() {
        WAE_REQUIRE(h && nsys >= 1 && coeffsA && coeffsM && v0 && H_out && V_out && m >= 1 && m <= 256, "bad argument");
        WAE_REQUIRE(op == WAE_OP_N || op == WAE_OP_C || op == WAE_OP_T, "bad op");
        require_solver(h);
        WAE_REQUIRE(nsys <= h->NB, "more systems than the solver batch width");
        return arnoldi_core(h, nsys, coeffsA, coeffsM, m, v0, -1, nullptr, op, tol, maxit, ritz_tol, H_out, V_out, info);
    });
}

int wae_arnoldi_shiftinvert(wae_family *h, const double *coeffsA, const double *coeffsM, int32_t m, const double *v0, int32_t op, double tol,
                            int32_t maxit, double *H_out, double *V_out, wae_solve_info *info) {
    return wae_arnoldi_shiftinvert_batch(h, 1, coeffsA, coeffsM, m, v0, op, tol, maxit, 0.0, H_out, V_out, info);
}

// wae_perturb (v0, v0adj, v_out in host memory) and wae_perturb_slots (v0d, v0adjd: device columns in the library's row numbering;
// v_out may be null: eigenvalue series only)
static int perturb_core(wae_family *h, const double *coeff_table, int32_t N, const double *v0, const double *v0adj, const cplx *v0d, const cplx *v0adjd,
                        int32_t norm_mode_in, const double *coeffsY, double tol, int32_t maxit, double *lambda_out, double *v_out, wae_solve_info *info) {
    {
        const bool skip_last = (norm_mode_in & 16) != 0;    // eigenvalue series only: no solve at order N
        const int norm_mode = norm_mode_in & 15;
        WAE_REQUIRE(norm_mode >= 0 && norm_mode <= 2 && (norm_mode != 2 || coeffsY), "bad norm_mode");
        require_solver(h);
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        wae_solve_info li;
        memset(&li, 0, sizeof(li));
        const double t0 = now_s();
        const int64_t d = h->d;
        const int T = h->T;
        auto F = [&](int m, int n, int k) { return zc(coeff_table[((size_t)(m * (N + 1) + n) * T + k) * 2], coeff_table[((size_t)(m * (N + 1) + n) * T + k) * 2 + 1]); };
        // (one allocation of the family, carved up and kept: twelve hipMalloc / hipFree pairs per call otherwise -- a Newton step of the
        // Householder iteration calls this once per start value)
        struct Span { cplx *p = nullptr; } PV, Ub, rb, rhs, u10, wl, tmp, tmp2, tmp3, sc;
        ensure(h->pt_ws, (size_t)d * (N + 1) + (size_t)d * T + (size_t)7 * d + 8);
        {
            cplx *q = h->pt_ws.p;
            PV.p = q; q += (size_t)d * (N + 1);
            Ub.p = q; q += (size_t)d * T;
            for (Span *sp : {&rb, &rhs, &u10, &wl, &tmp, &tmp2, &tmp3}) { sp->p = q; q += d; }
            sc.p = q;
        }
        DevBuf<cplx> &Gd = h->pt_Gd, &pcd = h->pt_pcd;
        ensure(Gd, (size_t)(N + 1) * T + 4);
        const OpDev A0 = h->ops[0].dev(WAE_OP_N);
        cplx hs[4];
        auto plane_tab = [&](const double *coeffs, int op) {     // level-0 plane table for an spmv
            std::vector<zc> pc;
            plane_coeffs(h, coeffs, op, pc);
            std::vector<cplx> tab(h->nplanes);
            for (int q = 0; q < h->nplanes; ++q) { const zc c = pc[h->slot_plane[0][q]]; tab[q] = cplx{c.real(), c.imag()}; }
            pcd.upload(tab.data(), tab.size(), st);
            HIP_CHECK(hipStreamSynchronize(st));
        };
        auto apply = [&](const double *coeffs, int op, const cplx *x, cplx *y) {
            plane_tab(coeffs, op);
            launch_spmv(h->ops[0].dev(op), pcd.p, 1 << 30, x, y, nullptr, 0.0, 1, MODE_AX, st);
            HIP_CHECK(hipStreamSynchronize(st));
        };
        auto dot = [&](const cplx *a, const cplx *b) -> zc {      // a^H b
            launch_dots(a, 0, 1, b, d, 1, h->partial.p, sc.p, st);
            HIP_CHECK(hipMemcpyAsync(hs, sc.p, sizeof(cplx), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            return zc(hs[0].x, hs[0].y);
        };
        auto axpby = [&](zc a, const cplx *x, zc b, const cplx *y, cplx *out) {   // out = a x + b y (out may alias x or y)
            cplx co[2] = {cplx{a.real(), a.imag()}, cplx{b.real(), b.imag()}};
            Gd.upload(co, 2, st);
            launch_lincomb(x, 0, 1, Gd.p, tmp2.p, d, 1, st);            // tmp2 = a x
            launch_lincomb(y, 0, 1, Gd.p + 1, tmp3.p, d, 1, st);        // tmp3 = b y
            launch_add(tmp2.p, tmp3.p, (size_t)d, st);
            launch_copy(tmp3.p, out, (size_t)d, st);
            HIP_CHECK(hipStreamSynchronize(st));
        };
        auto ipY = [&](const cplx *a, const cplx *b) -> zc {      // a^H Y b  (mode 2) or a^H b
            if (norm_mode != 2) return dot(a, b);
            apply(coeffsY, WAE_OP_N, b, tmp.p);
            return dot(a, tmp.p);
        };
        std::vector<double> c00(2 * T), c10(2 * T);
        for (int k = 0; k < T; ++k) {
            const zc a = F(0, 0, k), b = N >= 1 ? F(1, 0, k) : zc(0);
            c00[2 * k] = a.real(); c00[2 * k + 1] = a.imag();
            c10[2 * k] = b.real(); c10[2 * k + 1] = b.imag();
        }
        // v[0] = v0 / sqrt(ip(v0,v0))
        cplx *V0 = PV.p;
        if (v0d) {
            launch_copy(v0d, V0, (size_t)d, st);
            launch_copy(v0adjd, wl.p, (size_t)d, st);
        } else {
            ensure(h->io_a, (size_t)2 * d);                               // caller's row numbering -> the library's
            HIP_CHECK(hipMemcpyAsync(h->io_a.p, v0, (size_t)d * sizeof(cplx), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(h->io_a.p + d, v0adj, (size_t)d * sizeof(cplx), hipMemcpyHostToDevice, st));
            launch_colmajor_to_inter(h->io_a.p, d, 1, V0, 1, st, h->perm());
            launch_colmajor_to_inter(h->io_a.p + d, d, 1, wl.p, 1, st, h->perm());
        }
        {
            const zc nn = ipY(V0, V0);
            axpby(1.0 / std::sqrt(nn), V0, 0.0, V0, V0);
        }
        std::vector<zc> lam(N + 1, zc(0));
        if (N >= 1) {
            apply(c10.data(), WAE_OP_N, V0, u10.p);                                  // u10 = L(1,0) v0
            Batch bt;
            bt.nb = 1; bt.cps = 1; bt.nsys = 1; bt.op = WAE_OP_N;
            std::vector<std::vector<zc>> pcs(1);
            if (norm_mode == 2) {                                                    // perturbation.jl:493-494
                plane_coeffs(h, coeffsY, WAE_OP_N, pcs[0]);
                solve_chunk(h, bt, pcs, wl.p, tmp.p, tol, maxit, &li);               // v0Adj = Y \ v0Adj
                apply(coeffsY, WAE_OP_N, u10.p, tmp2.p);
                const zc sN = dot(tmp.p, tmp2.p);                                    // v0Adj' Y L10 v0
                launch_copy(tmp.p, rb.p, (size_t)d, st);
                axpby(1.0 / sN, rb.p, 0.0, rb.p, rb.p);                              // v0Adj /= s
                apply(coeffsY, WAE_OP_C, rb.p, wl.p);                                // wl = Y' v0Adj
            } else {
                const zc sN = dot(wl.p, u10.p);                                      // v0Adj' L10 v0
                axpby(1.0 / sN, wl.p, 0.0, wl.p, wl.p);
            }
            const zc denom = dot(wl.p, u10.p);
            plane_coeffs(h, c00.data(), WAE_OP_N, pcs[0]);
            bool l00_ready = false;      // set up lazily: order-1 Newton steps (skip_last) never solve with L(0,0), which is
                                         // exactly singular for small dense families (the reference would throw there)
            // plane passes for the multi-input SpMV (coefficient 1 per term: the weights live in G)
            std::vector<std::vector<int>> plane_terms(h->nplanes);
            for (int k = 0; k < T; ++k) plane_terms[h->term_plane[k]].push_back(k);
            size_t npass = 0;
            for (auto &v : plane_terms) npass = std::max(npass, v.size());
            std::vector<zc> G;
            for (int k = 1; k <= N; ++k) {
                G.assign((size_t)k * T, zc(0));
                auto addF = [&](int i, int m, int n, zc coeff) {
                    for (int t = 0; t < T; ++t) G[(size_t)i * T + t] += coeff * F(m, n, t);
                };
                for (int n = 1; n <= k; ++n) addF(k - n, 0, n, 1.0);
                for (int mw = 1; mw <= k; ++mw)
                    for_each_partition(mw, [&](const int *p, int len) {
                        if (len == 1 && p[0] == k) return;
                        std::vector<int> mu(mw, 0);
                        for (int i = 0; i < len; ++i) mu[p[i] - 1]++;
                        double mn = std::tgamma((double)len + 1.0);
                        zc coeff = 1.0;
                        for (int g = 0; g < mw; ++g)
                            if (mu[g]) {
                                mn /= std::tgamma((double)mu[g] + 1.0);
                                coeff *= std::pow(lam[g + 1], mu[g]);
                            }
                        coeff *= mn;
                        for (int n = 0; n <= k - mw; ++n) {
                            if (k == 1 && len == 1) continue;
                            addF(k - n - mw, len, n, coeff);
                        }
                    });
                std::vector<cplx> Gc((size_t)k * T);
                for (size_t i = 0; i < Gc.size(); ++i) Gc[i] = cplx{G[i].real(), G[i].imag()};
                Gd.upload(Gc.data(), Gc.size(), st);
                launch_gemv_multi(PV.p, (size_t)d, k, Gd.p, Ub.p, d, T, st);
                HIP_CHECK(hipStreamSynchronize(st));
                launch_fill_zero(rb.p, (size_t)d, st);
                std::vector<cplx> tab(h->nplanes);
                std::vector<int> pcol(h->nplanes);
                for (size_t ps = 0; ps < npass; ++ps) {
                    for (int sidx = 0; sidx < h->nplanes; ++sidx) {
                        const int q = h->slot_plane[0][sidx];
                        if (ps < plane_terms[q].size()) {
                            const int kk = plane_terms[q][ps];
                            const zc c = h->term_scale[kk];
                            tab[sidx] = cplx{c.real(), c.imag()};
                            pcol[sidx] = kk;
                        } else { tab[sidx] = cplx{0.0, 0.0}; pcol[sidx] = 0; }
                    }
                    pcd.upload(tab.data(), tab.size(), st);
                    h->plane_col_dev.upload(pcol.data(), pcol.size(), st);
                    launch_spmv_multi(A0, pcd.p, h->plane_col_dev.p, Ub.p, tmp.p, T, st);
                    launch_add(tmp.p, rb.p, (size_t)d, st);
                    HIP_CHECK(hipStreamSynchronize(st));
                }
                lam[k] = -dot(wl.p, rb.p) / denom;
                if (skip_last && k == N) break;
                axpby(-1.0, rb.p, -lam[k], u10.p, rhs.p);                             // rhs = -(r + lam_k L10 v0)
                cplx *vk = PV.p + (size_t)k * d;
                if (!l00_ready) { upload_pc(h, pcs); dense_setup(h, bt); l00_ready = true; }
                gmres(h, bt, rhs.p, vk, tol, maxit, &li);
                const zc pr = ipY(V0, vk);
                axpby(1.0, vk, -pr, V0, vk);                                          // v_k -= (v0' [Y] v_k) v0
                if (norm_mode >= 1) {
                    zc c = 0;
                    for (int l = 1; l < k; ++l) c -= 0.5 * ipY(PV.p + (size_t)l * d, PV.p + (size_t)(k - l) * d);
                    axpby(1.0, vk, c, V0, vk);
                }
            }
        }
        if (v_out) {
            ensure(h->io_b, (size_t)d * (N + 1));
            for (int k = 0; k <= N; ++k) launch_inter_to_colmajor(PV.p + (size_t)k * d, 1, d, 1, h->io_b.p + (size_t)k * d, st, h->perm());
            HIP_CHECK(hipMemcpyAsync(v_out, h->io_b.p, (size_t)d * (N + 1) * sizeof(cplx), hipMemcpyDeviceToHost, st));
        }
        HIP_CHECK(hipStreamSynchronize(st));
        for (int k = 1; k <= N; ++k) { lambda_out[2 * k] = lam[k].real(); lambda_out[2 * k + 1] = lam[k].imag(); }
        li.seconds = now_s() - t0;
        const int rc_ = info_code(li);
        if (info) *info = li;
        return rc_;
    }
}

int wae_perturb(wae_family *h, const double *coeff_table, int32_t N, const double *v0, const double *v0adj, int32_t norm_mode_in, const double *coeffsY,
                double tol, int32_t maxit, double *lambda_out, double *v_out, wae_solve_info *info) {
    return guarded([&]() {
        WAE_REQUIRE(h && coeff_table && v0 && v0adj && lambda_out && v_out && N >= 0 && N <= 200, "bad argument");
        return perturb_core(h, coeff_table, N, v0, v0adj, nullptr, nullptr, norm_mode_in, coeffsY, tol, maxit, lambda_out, v_out, info);
    });
}

// ----------------------------------------------------------------------------------------------------
// Device-resident multivectors ("slots").  The Newton-type solvers iterate on a handful of vectors per start value (right and left
// eigenvector estimates, the Ritz vectors of the step): with the processes' inputs and outputs in host memory a Householder step of 8
// start values at 1M DoF moved 2 GB over PCIe into freshly touched pages and spent as long in host copies as the GPU spent computing
// (45 % idle in the kernel trace).  A slot holds d x ncols in the library's row numbering; the calls below read and write slot columns
// in place of host arrays, so that an iteration touches host memory once at its start and once at its end.
// ----------------------------------------------------------------------------------------------------
static wae_family::Slot &slot_ref(wae_family *h, int32_t slot) {
    WAE_REQUIRE(slot >= 0 && slot < WAE_NSLOTS, "slot index out of range");
    return h->slots[slot];
}
static cplx *slot_col(wae_family *h, int32_t slot, int32_t col) {
    wae_family::Slot &S = slot_ref(h, slot);
    WAE_REQUIRE(S.ncols > 0, "slot is empty");
    WAE_REQUIRE(col >= 0 && col < S.ncols, "slot column out of range");
    return S.buf.p + (size_t)col * h->d;
}
// n columns of a slot as one contiguous column-major block: the slot's own memory if they are consecutive, a copy in `stage` otherwise
static const cplx *slot_cols_ptr(wae_family *h, int32_t slot, const int32_t *cols, int n, DevBuf<cplx> &stage, hipStream_t st) {
    WAE_REQUIRE(cols && n >= 1, "bad argument");
    bool consecutive = true;
    for (int i = 0; i < n; ++i) { (void)slot_col(h, slot, cols[i]); consecutive = consecutive && cols[i] == cols[0] + i; }
    if (consecutive) return slot_col(h, slot, cols[0]);
    ensure(stage, (size_t)n * h->d);
    for (int i = 0; i < n; ++i) launch_copy(slot_col(h, slot, cols[i]), stage.p + (size_t)i * h->d, (size_t)h->d, st);
    return stage.p;
}

int wae_slot_write(wae_family *h, int32_t slot, int32_t ncols_total, int32_t col0, int32_t ncols, const double *X) {
    return guarded([&]() {
        WAE_REQUIRE(h && ncols_total >= 1 && ncols_total <= 256 && col0 >= 0 && ncols >= 0 && col0 + ncols <= ncols_total && (ncols == 0 || X), "bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        wae_family::Slot &S = slot_ref(h, slot);
        const int64_t d = h->d;
        if (S.ncols != ncols_total) {                        // (re)created: zero columns
            S.buf.alloc((size_t)d * ncols_total);
            S.ncols = ncols_total;
            launch_fill_zero(S.buf.p, (size_t)d * ncols_total, st);
        }
        if (ncols) {
            ensure(h->io_a, (size_t)d * ncols);
            HIP_CHECK(hipMemcpyAsync(h->io_a.p, X, (size_t)d * ncols * sizeof(cplx), hipMemcpyHostToDevice, st));
            for (int c = 0; c < ncols; ++c) launch_colmajor_to_inter(h->io_a.p + (size_t)c * d, d, 1, S.buf.p + (size_t)(col0 + c) * d, 1, st, h->perm());
        }
        HIP_CHECK(hipStreamSynchronize(st));
        return WAE_OK;
    });
}

int wae_slot_read(wae_family *h, int32_t slot, int32_t col0, int32_t ncols, double *X) {
    return guarded([&]() {
        WAE_REQUIRE(h && col0 >= 0 && ncols >= 0 && (ncols == 0 || X), "bad argument");
        if (!ncols) return WAE_OK;
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        (void)slot_col(h, slot, col0);
        (void)slot_col(h, slot, col0 + ncols - 1);
        const int64_t d = h->d;
        ensure(h->io_b, (size_t)d * ncols);
        for (int c = 0; c < ncols; ++c) launch_inter_to_colmajor(slot_col(h, slot, col0 + c), 1, d, 1, h->io_b.p + (size_t)c * d, st, h->perm());
        HIP_CHECK(hipMemcpyAsync(X, h->io_b.p, (size_t)d * ncols * sizeof(cplx), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        return WAE_OK;
    });
}

int wae_slot_axpby(wae_family *h, int32_t n, int32_t dst_slot, const int32_t *dst_cols, int32_t src_slot, const int32_t *src_cols, const double *alpha,
                   const double *beta, int32_t conj_src) {
    return guarded([&]() {
        WAE_REQUIRE(h && n >= 0 && (n == 0 || (dst_cols && src_cols && alpha && beta)), "bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        for (int i = 0; i < n; ++i) {
            cplx *y = slot_col(h, dst_slot, dst_cols[i]);
            const cplx *x = slot_col(h, src_slot, src_cols[i]);
            launch_axpby1(cplx{alpha[2 * i], alpha[2 * i + 1]}, x, cplx{beta[2 * i], beta[2 * i + 1]}, y, y, (size_t)h->d, st, conj_src ? 1 : 0);
        }
        HIP_CHECK(hipStreamSynchronize(st));
        return WAE_OK;
    });
}

int wae_slot_forms(wae_family *h, int32_t n, const double *coeffs, int32_t op, int32_t a_slot, const int32_t *a_cols, int32_t b_slot, const int32_t *b_cols,
                   double *out) {
    return guarded([&]() {
        WAE_REQUIRE(h && n >= 0 && (n == 0 || (coeffs && a_cols && b_cols && out)), "bad argument");
        WAE_REQUIRE(op == WAE_OP_N || op == WAE_OP_C || op == WAE_OP_T, "bad op");
        if (!n) return WAE_OK;
        WAE_REQUIRE(n <= 256, "more than 256 forms in one call");
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        const int64_t d = h->d;
        const int T = h->T;
        const size_t vec = (size_t)d * n;
        ensure(h->io_b, 3 * vec);
        cplx *Bi = h->io_b.p, *ABi = Bi + vec, *Ai = ABi + vec;
        launch_colmajor_to_inter(slot_cols_ptr(h, b_slot, b_cols, n, h->io_a, st), d, n, Bi, n, st, nullptr);
        launch_colmajor_to_inter(slot_cols_ptr(h, a_slot, a_cols, n, h->io_a, st), d, n, Ai, n, st, nullptr);
        std::vector<cplx> tab((size_t)n * h->nplanes);
        std::vector<zc> pc;
        for (int i = 0; i < n; ++i) {
            plane_coeffs(h, coeffs + (size_t)i * 2 * T, op, pc);
            for (int q = 0; q < h->nplanes; ++q) { const zc c = pc[h->slot_plane[0][q]]; tab[(size_t)i * h->nplanes + q] = cplx{c.real(), c.imag()}; }
        }
        h->pt_pcd.upload(tab.data(), tab.size(), st);
        launch_spmv(h->ops[0].dev(op), h->pt_pcd.p, 1, Bi, ABi, nullptr, 0.0, n, MODE_AX, st);
        ensure(h->partial, (size_t)1024 * 32 * std::max(n, 8));      // (launch_dots: DOT_BLOCKS x vectors x columns; the solver set-up allocates more)
        ensure(h->pt_Gd, (size_t)n);
        launch_dots(Ai, 0, 1, ABi, d, n, h->partial.p, h->pt_Gd.p, st);
        std::vector<cplx> r(n);
        HIP_CHECK(hipMemcpyAsync(r.data(), h->pt_Gd.p, (size_t)n * sizeof(cplx), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        for (int i = 0; i < n; ++i) { out[2 * i] = r[i].x; out[2 * i + 1] = r[i].y; }
        return WAE_OK;
    });
}

int wae_arnoldi_shiftinvert_slots(wae_family *h, int32_t nsys, const double *coeffsA, const double *coeffsM, int32_t m, int32_t v0_slot,
                                  const int32_t *v0_cols, int32_t op, double tol, int32_t maxit, double ritz_tol, double *H_out, wae_solve_info *info) {
    return guarded([&]() {
        WAE_REQUIRE(h && nsys >= 1 && coeffsA && coeffsM && v0_cols && H_out && m >= 1 && m <= 256, "bad argument");
        WAE_REQUIRE(op == WAE_OP_N || op == WAE_OP_C || op == WAE_OP_T, "bad op");
        require_solver(h);
        WAE_REQUIRE(nsys <= h->NB, "more systems than the solver batch width");
        for (int i = 0; i < nsys; ++i) (void)slot_col(h, v0_slot, v0_cols[i]);
        return arnoldi_core(h, nsys, coeffsA, coeffsM, m, nullptr, v0_slot, v0_cols, op, tol, maxit, ritz_tol, H_out, nullptr, info);
    });
}

int wae_arnoldi_ritz_to_slot(wae_family *h, int32_t nsys, int32_t ny, const double *y, int32_t dst_slot, const int32_t *dst_cols, int32_t normalise) {
    return guarded([&]() {
        WAE_REQUIRE(h && nsys >= 1 && ny >= 1 && y && dst_cols, "bad argument");
        WAE_REQUIRE(h->arn_nsys == nsys && ny <= h->arn_cols, "no Arnoldi basis of that shape on the device (wae_arnoldi_shiftinvert_slots first)");
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        const int64_t d = h->d;
        const size_t vec = (size_t)d * nsys;
        for (int i = 0; i < nsys; ++i) (void)slot_col(h, dst_slot, dst_cols[i]);
        std::vector<cplx> yc((size_t)ny * nsys);
        for (int sy = 0; sy < nsys; ++sy)
            for (int j = 0; j < ny; ++j) yc[(size_t)j * nsys + sy] = cplx{y[((size_t)sy * ny + j) * 2], y[((size_t)sy * ny + j) * 2 + 1]};
        ensure(h->arn_hcol, yc.size() + (size_t)nsys);
        HIP_CHECK(hipMemcpyAsync(h->arn_hcol.p, yc.data(), yc.size() * sizeof(cplx), hipMemcpyHostToDevice, st));
        launch_lincomb(h->arn_EV.p, vec, ny, h->arn_hcol.p, h->arn_t.p, d, nsys, st);
        if (normalise) {
            launch_norms(h->arn_t.p, d, nsys, h->partial.p, h->arn_hcol.p + yc.size(), st);
            launch_scale_inv(h->arn_t.p, h->arn_hcol.p + yc.size(), h->arn_t.p, d, nsys, st);
        }
        launch_inter_to_colmajor(h->arn_t.p, nsys, d, nsys, h->arn_stage.p, st, nullptr);
        for (int i = 0; i < nsys; ++i) launch_copy(h->arn_stage.p + (size_t)i * d, slot_col(h, dst_slot, dst_cols[i]), (size_t)d, st);
        HIP_CHECK(hipStreamSynchronize(st));                 // (yc is a stack vector)
        return WAE_OK;
    });
}

int wae_perturb_slots(wae_family *h, const double *coeff_table, int32_t N, int32_t v_slot, int32_t v_col, int32_t vadj_slot, int32_t vadj_col,
                      int32_t norm_mode_in, const double *coeffsY, double tol, int32_t maxit, double *lambda_out, double *v_out, wae_solve_info *info) {
    return guarded([&]() {
        WAE_REQUIRE(h && coeff_table && lambda_out && N >= 0 && N <= 200, "bad argument");
        return perturb_core(h, coeff_table, N, nullptr, nullptr, slot_col(h, v_slot, v_col), slot_col(h, vadj_slot, vadj_col), norm_mode_in, coeffsY,
                            tol, maxit, lambda_out, v_out, info);
    });
}

int wae_bench_spmv(wae_family *h, const double *coeffs, int32_t r, int32_t reps, double *ms_out) {
    return guarded([&]() {
        WAE_REQUIRE(h && coeffs && r > 0 && reps > 0 && ms_out, "bad argument");
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        const size_t cnt = (size_t)h->d * r;
        DevBuf<cplx> x, y, pcd;
        x.alloc(cnt); y.alloc(cnt);
        std::vector<cplx> hx(cnt);
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (size_t i = 0; i < cnt; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            hx[i].x = (double)(s & 0xFFFFF) / 524288.0 - 1.0;
            hx[i].y = (double)((s >> 20) & 0xFFFFF) / 524288.0 - 1.0;
        }
        HIP_CHECK(hipMemcpyAsync(x.p, hx.data(), cnt * sizeof(cplx), hipMemcpyHostToDevice, st));
        std::vector<zc> pc;
        plane_coeffs(h, coeffs, WAE_OP_N, pc);
        // WAE_BENCH_CPS = columns per system (diagnostic: 1, 2 or 4 is what a rank of a multi-GPU pass sees: its share of the
        // probe columns of every system); default: one system, all columns
        const int cps = getenv("WAE_BENCH_CPS") ? std::max(1, atoi(getenv("WAE_BENCH_CPS"))) : (1 << 30);
        const int nsys = cps >= r ? 1 : (r + cps - 1) / cps;
        std::vector<cplx> tab((size_t)h->nplanes * nsys);
        for (int sidx = 0; sidx < nsys; ++sidx)
            for (int q = 0; q < h->nplanes; ++q) { const zc c = pc[h->slot_plane[0][q]]; tab[(size_t)sidx * h->nplanes + q] = cplx{c.real(), c.imag()}; }
        pcd.upload(tab.data(), tab.size(), st);
        const OpDev A = h->ops[0].dev(WAE_OP_N);
        // WAE_BENCH_MODE (diagnostic): the fused form timed -- 0 A X (default), 1 residual, 2 Jacobi sweep, 6 product + first sweep
        const int bmode = getenv("WAE_BENCH_MODE") ? atoi(getenv("WAE_BENCH_MODE")) : MODE_AX;
        DevBuf<cplx> bb;
        if (bmode != MODE_AX) { bb.alloc(cnt); HIP_CHECK(hipMemcpyAsync(bb.p, hx.data(), cnt * sizeof(cplx), hipMemcpyHostToDevice, st)); }
        auto one = [&]() { launch_spmv(A, pcd.p, cps, x.p, y.p, bmode != MODE_AX ? bb.p : nullptr, 0.8, r, bmode, st); };
        for (int i = 0; i < 3; ++i) one();
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) one();
        HIP_CHECK(hipEventRecord(e1, st));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = (double)ms / reps;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        x.release(); y.release(); pcd.release();
        return WAE_OK;
    });
}

int wae_debug_spmv(wae_family *h, int32_t which, int32_t level, int32_t mode, const double *coeffs, int32_t ncoef, const double *X,
                   const double *B, double *Y, double *B2, int32_t r, int32_t op, double jac_w, const uint8_t *cmask, int32_t flags,
                   int64_t *n_in_q, int64_t *n_out_q) {
    return guarded([&]() {
        WAE_REQUIRE(h && which >= 0 && which <= 2 && level >= 0 && op >= 0 && op <= 2, "bad argument");
        WAE_REQUIRE(level == 0 || h->solver_ready, "levels >= 1 need wae_solver_setup");
        // (the last level of a hierarchy is dense: it has no sparse operator to launch)
        WAE_REQUIRE(which == 0 ? (level == 0 || level < (int)h->ops.size() - 1) : level < (int)h->xfer.size(), "no such level");
        const int64_t n_in = which == 0 ? h->ops[level].n : (which == 1 ? h->xfer[level].nf : h->xfer[level].nc);
        const int64_t n_out = which == 0 ? h->ops[level].n : (which == 1 ? h->xfer[level].nc : h->xfer[level].nf);
        if (n_in_q) *n_in_q = n_in;
        if (n_out_q) *n_out_q = n_out;
        if (!X && !Y) return WAE_OK;                               // size query
        WAE_REQUIRE(X && Y && r > 0 && r <= 256, "bad argument (1 <= r <= 256)");
        WAE_REQUIRE(which != 0 || (coeffs && (ncoef == 1 || ncoef == r)), "ncoef must be 1 or r");
        WAE_REQUIRE(mode >= MODE_AX && mode <= MODE_AX_J0 && (which == 0 || mode == (which == 1 ? MODE_AX : MODE_ADD)), "bad mode");
        WAE_REQUIRE(B || mode == MODE_AX || mode == MODE_AX_DS || mode == MODE_AX_J0, "this mode reads B");
        WAE_REQUIRE(mode != MODE_AX_J0 || B2, "mode 6 writes B2");
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        const int *perm = level == 0 && which != 1 ? h->perm() : nullptr;                 // (numbering of the OUTPUT rows: level 0 is the caller's)
        const int *perm_in = level == 0 && which != 2 ? h->perm() : nullptr;
        DevBuf<cplx> xc, yc, xi, yi, bi, pcd;
        DevBuf<unsigned char> cm;
        const size_t cin = (size_t)n_in * r, cout = (size_t)n_out * r;
        xc.alloc(std::max(cin, cout)); yc.alloc(cout); xi.alloc(cin); yi.alloc(cout); bi.alloc(cout);
        HIP_CHECK(hipMemcpyAsync(xc.p, X, cin * sizeof(cplx), hipMemcpyHostToDevice, st));
        launch_colmajor_to_inter(xc.p, n_in, r, xi.p, r, st, perm_in);
        auto stage = [&](const double *src, cplx *dst) {           // host column-major n_out x r -> interleaved on the device
            HIP_CHECK(hipMemcpyAsync(yc.p, src, cout * sizeof(cplx), hipMemcpyHostToDevice, st));
            launch_colmajor_to_inter(yc.p, n_out, r, dst, r, st, perm);
        };
        stage(Y, yi.p);                                             // a masked chunk keeps what Y held on entry
        if (mode == MODE_AX_J0) stage(B2, bi.p);
        else if (B) stage(B, bi.p);
        if (cmask) {
            cm.alloc((size_t)(r + 7) / 8);
            HIP_CHECK(hipMemcpyAsync(cm.p, cmask, (size_t)(r + 7) / 8, hipMemcpyHostToDevice, st));
        }
        OpDev A;
        int cps = 1 << 30;
        if (which == 0) {
            std::vector<cplx> tab((size_t)ncoef * h->nplanes);
            std::vector<zc> pc;
            for (int s = 0; s < ncoef; ++s) {
                plane_coeffs(h, coeffs + (size_t)s * 2 * h->T, op, pc);
                for (int q = 0; q < h->nplanes; ++q) { const zc c = pc[h->slot_plane[level][q]]; tab[(size_t)s * h->nplanes + q] = cplx{c.real(), c.imag()}; }
            }
            pcd.upload(tab.data(), tab.size(), st);
            A = h->ops[level].dev(op);
            cps = ncoef == 1 ? (1 << 30) : 1;
        } else {
            A = h->xfer[level].devR();
        }
        if (flags & 1) A.tiles = nullptr;
        const Transfer *xf = which == 0 ? nullptr : &h->xfer[level];
        const bool by_tile = xf && xf->ft.ready && !(flags & 1) && r >= 8;
        if (which == 2) {                                           // Y = B + P X, in place on the staged B
            launch_copy(bi.p, yi.p, cout, st);                      // (masked chunks: Y and B agree there only if the caller passed them equal)
            if (by_tile) launch_prolong_tiles(xf->ft.dev, xi.p, yi.p, r, st, cmask ? cm.p : nullptr);
            else launch_prolong_add(xf->p_ptr.p, xf->p_col.p, xf->p_val.p, xf->nf, xi.p, yi.p, r, st, cmask ? cm.p : nullptr);
        } else
        launch_spmv(A, which == 0 ? pcd.p : h->one_dev.p, cps, xi.p, yi.p, (mode == MODE_AX || mode == MODE_AX_DS) ? nullptr : bi.p, jac_w, r, mode, st,
                    cmask ? cm.p : nullptr);
        launch_inter_to_colmajor(yi.p, r, n_out, r, yc.p, st, perm);
        HIP_CHECK(hipMemcpyAsync(Y, yc.p, cout * sizeof(cplx), hipMemcpyDeviceToHost, st));
        if (mode == MODE_AX_J0) {
            HIP_CHECK(hipStreamSynchronize(st));
            launch_inter_to_colmajor(bi.p, r, n_out, r, yc.p, st, perm);
            HIP_CHECK(hipMemcpyAsync(B2, yc.p, cout * sizeof(cplx), hipMemcpyDeviceToHost, st));
        }
        HIP_CHECK(hipStreamSynchronize(st));
        return WAE_OK;
    });
}

int wae_bench_spmv_level(wae_family *h, const double *coeffs, int32_t which, int32_t level, int32_t r, int32_t reps, double *ms_out,
                         int64_t *bytes_out) {
    return guarded([&]() {
        WAE_REQUIRE(h && coeffs && which >= 0 && which <= 2 && level >= 0 && r > 0 && r <= 256 && reps > 0 && ms_out, "bad argument");
        require_solver(h);
        WAE_REQUIRE(which == 0 ? (level == 0 || level < (int)h->ops.size() - 1) : level < (int)h->xfer.size(), "no such level");
        HIP_CHECK(hipSetDevice(h->device));
        hipStream_t st = h->stream;
        const int64_t n_in = which == 0 ? h->ops[level].n : (which == 1 ? h->xfer[level].nf : h->xfer[level].nc);
        const int64_t n_out = which == 0 ? h->ops[level].n : (which == 1 ? h->xfer[level].nc : h->xfer[level].nf);
        DevBuf<cplx> x, y, pcd;
        x.alloc((size_t)n_in * r); y.alloc((size_t)n_out * r);
        std::vector<cplx> hx((size_t)n_in * r);
        uint64_t sd = 0x9E3779B97F4A7C15ull;
        for (auto &v : hx) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; v.x = (double)(sd & 0xFFFFF) / 524288.0 - 1.0; v.y = (double)((sd >> 20) & 0xFFFFF) / 524288.0 - 1.0; }
        HIP_CHECK(hipMemcpyAsync(x.p, hx.data(), hx.size() * sizeof(cplx), hipMemcpyHostToDevice, st));
        OpDev A;
        const cplx *pcp = h->one_dev.p;
        int64_t bytes = 2;
        if (which == 0) {
            std::vector<zc> pc;
            plane_coeffs(h, coeffs, WAE_OP_N, pc);
            std::vector<cplx> tab(h->nplanes);
            bytes = 0;
            for (int q = 0; q < h->nplanes; ++q) {
                const zc c = pc[h->slot_plane[level][q]];
                tab[q] = cplx{c.real(), c.imag()};
            }
            pcd.upload(tab.data(), tab.size(), st);
            pcp = pcd.p;
            A = h->ops[level].dev(WAE_OP_N);
            bytes = 0;
            for (size_t g = 0; g < h->ops[level].groups.size(); ++g) {
                const GroupHost &G = h->ops[level].groups[g];
                for (int q = 0; q < G.nplanes; ++q)
                    if (tab[G.plane0 + q].x != 0.0 || tab[G.plane0 + q].y != 0.0) bytes += G.nnz * 20 + (n_out + 1) * 4;
            }
        } else {
            A = h->xfer[level].devR();
            bytes = (int64_t)h->xfer[level].r_col.n * 12 + (n_out + 1) * 4;       // real values: 8 + 4 bytes per entry
        }
        bytes += (int64_t)r * (n_in + n_out) * 16;
        if (which == 2) bytes += (int64_t)r * n_out * 16;          // (the prolongation updates the fine vector in place: read + write)
        if (bytes_out) *bytes_out = bytes;
        const Transfer *xf = which == 0 ? nullptr : &h->xfer[level];
        const bool by_tile = xf && xf->ft.ready && r >= 8 && xfer_tiles_on();
        auto one = [&]() {
            if (which == 2) {
                if (by_tile) launch_prolong_tiles(xf->ft.dev, x.p, y.p, r, st);
                else launch_prolong_add(xf->p_ptr.p, xf->p_col.p, xf->p_val.p, xf->nf, x.p, y.p, r, st, nullptr);
            } else launch_spmv(A, pcp, 1 << 30, x.p, y.p, nullptr, 0.0, r, MODE_AX, st);
        };
        if (which == 2) HIP_CHECK(hipMemsetAsync(y.p, 0, (size_t)n_out * r * sizeof(cplx), st));
        for (int i = 0; i < 3; ++i) one();
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) one();
        HIP_CHECK(hipEventRecord(e1, st));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = (double)ms / reps;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return WAE_OK;
    });
}

int wae_bench_triad(int32_t device, int64_t n, int32_t reps, double *gbs_out) {
    return guarded([&]() {
        WAE_REQUIRE(n > 0 && reps > 0 && gbs_out, "bad argument");
        HIP_CHECK(hipSetDevice(device));
        DevBuf<double> a, b, c;
        a.alloc(n); b.alloc(n); c.alloc(n);
        HIP_CHECK(hipMemset(b.p, 0, n * sizeof(double)));
        HIP_CHECK(hipMemset(c.p, 0, n * sizeof(double)));
        hipStream_t st;
        HIP_CHECK(hipStreamCreate(&st));
        // The rate depends on the shape of the launch (measured, round 4: 4.6 ... 5.7 TB/s between 256 and 65 536 workgroups on one
        // box; the 8 192 this probe used until then sits at the low end): the best of five grid sizes is what the device attains.
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        double best = 0.0;
        for (unsigned cap : {512u, 1024u, 4096u, 8192u, 65536u}) {
            for (int i = 0; i < 2; ++i) launch_triad(a.p, b.p, c.p, 1.5, n, st, cap);
            HIP_CHECK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) launch_triad(a.p, b.p, c.p, 1.5, n, st, cap);
            HIP_CHECK(hipEventRecord(e1, st));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::max(best, 3.0 * n * sizeof(double) * reps / (ms * 1e-3) / 1e9);
        }
        *gbs_out = best;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipStreamDestroy(st);
        a.release(); b.release(); c.release();
        return WAE_OK;
    });
}

}   // extern "C"
