// Host-side sparse helpers and smoothed-aggregation set-up for the multigrid preconditioner.
// (Set-up runs once per family on the host cores; everything it produces lives in HBM afterwards.)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>

#include "amg.h"
#include "hugemem.h"

int g_amg_fused_prolongator = 1;     // (test hook, amg.h)
#include <chrono>
#include <cstdlib>
#include <functional>
#include <future>
#include <thread>

template <class V> static void transpose_impl(int64_t n, int64_t m, const std::vector<int> &ptr, const std::vector<int> &col,
                                              const std::vector<V> &val, std::vector<int> &tptr, std::vector<int> &tcol, std::vector<V> &tval) {
    tptr.assign(m + 1, 0);
    for (int c : col) tptr[c + 1]++;
    for (int64_t i = 0; i < m; ++i) tptr[i + 1] += tptr[i];
    tcol.resize(col.size());
    tval.resize(val.size());
    std::vector<int> pos(tptr.begin(), tptr.end() - 1);
    for (int64_t i = 0; i < n; ++i)
        for (int p = ptr[i]; p < ptr[i + 1]; ++p) {
            int q = pos[col[p]]++;
            tcol[q] = (int)i;
            tval[q] = val[p];
        }
}
CsrZ csr_transpose(const CsrZ &A) {
    CsrZ T;
    T.n = A.m; T.m = A.n;
    transpose_impl(A.n, A.m, A.ptr, A.col, A.val, T.ptr, T.col, T.val);
    return T;
}
CsrD csr_transpose(const CsrD &A) {
    CsrD T;
    T.n = A.m; T.m = A.n;
    transpose_impl(A.n, A.m, A.ptr, A.col, A.val, T.ptr, T.col, T.val);
    return T;
}
bool csr_same_pattern(const CsrZ &A, const CsrZ &B) { return A.n == B.n && A.m == B.m && A.ptr == B.ptr && A.col == B.col; }

// Host threads for the set-up (the set-up is part of the metric: one cold solver call = set-up + pass).  32 by default: the phases
// are bound by memory stalls and page faults, not by arithmetic, and on a 16-CPU share 24..64 threads all finish the 1M-DoF set-up
// in 0.93-1.05 s where 16 take 1.22-1.31 and 8 take 1.5-1.9 (the kernel's CPU quota does the limiting).  The row loops below are
// cut into contiguous chunks, one per thread, each with its own marker / accumulator and output vectors, stitched afterwards:
// the result is the serial one bit for bit (per row the same operations in the same order).
static int setup_threads() {
    static const int n = []() {
        if (const char *e = getenv("WAE_SETUP_THREADS")) return std::max(1, atoi(e));
        return (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    }();
    return n;
}
// runs body(lo, hi, part) over nparts contiguous row ranges of [0, n) on as many threads
template <class F> static void parallel_ranges(int64_t n, int nparts, F &&body) {
    nparts = (int)std::max<int64_t>(1, std::min<int64_t>(nparts, n / 4096 + 1));
    if (nparts == 1) { body((int64_t)0, n, 0); return; }
    std::vector<std::future<void>> jobs;
    for (int t = 0; t < nparts; ++t) {
        const int64_t lo = n * t / nparts, hi = n * (t + 1) / nparts;
        jobs.push_back(std::async(std::launch::async, [&body, lo, hi, t]() { body(lo, hi, t); }));
    }
    for (auto &j : jobs) j.get();
}
// stitches per-part (col, val) pieces; ptr holds per-row counts in ptr[i+1] on entry
template <class V> static void stitch(int64_t n, int nparts, std::vector<int> &ptr, std::vector<std::vector<int>> &pcol, std::vector<std::vector<V>> &pval,
                                      std::vector<int> &col, std::vector<V> &val) {
    (void)nparts;
    for (int64_t i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
    huge_reserve(col, (size_t)ptr[n]); huge_reserve(val, (size_t)ptr[n]);
    col.resize((size_t)ptr[n]);
    val.resize((size_t)ptr[n]);
    // every part copies its own piece (the serial copy of 300 MB per product was a tenth of a second each at 1M unknowns)
    std::vector<size_t> off(pcol.size() + 1, 0);
    for (size_t t = 0; t < pcol.size(); ++t) off[t + 1] = off[t] + pcol[t].size();
    std::vector<std::future<void>> jobs;
    for (size_t t = 0; t < pcol.size(); ++t)
        jobs.push_back(std::async(std::launch::async, [&, t]() {
            std::copy(pcol[t].begin(), pcol[t].end(), col.begin() + off[t]);
            std::copy(pval[t].begin(), pval[t].end(), val.begin() + off[t]);
            std::vector<int>().swap(pcol[t]);
            std::vector<V>().swap(pval[t]);
        }));
    for (auto &j : jobs) j.get();
}

// C = A * B   (row-wise Gustavson with a dense marker; columns sorted; structural zeros kept so that planes
// with equal patterns keep equal patterns after projection)
template <class VA, class VB, class VC>
static void spgemm(int64_t n, int64_t m, const std::vector<int> &aptr, const std::vector<int> &acol, const std::vector<VA> &aval,
                   const std::vector<int> &bptr, const std::vector<int> &bcol, const std::vector<VB> &bval, std::vector<int> &cptr,
                   std::vector<int> &ccol, std::vector<VC> &cval, int nthreads = 1) {
    cptr.assign(n + 1, 0);
    const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, n / 4096 + 1));
    std::vector<std::vector<int>> pcol(nparts);
    std::vector<std::vector<VC>> pval(nparts);
    parallel_ranges(n, nparts, [&](int64_t lo, int64_t hi, int part) {
        std::vector<int> marker(m, -1);
        std::vector<VC> acc(m);
        std::vector<int> list;
        std::vector<int> &oc = pcol[part];
        std::vector<VC> &ov = pval[part];
        for (int64_t i = lo; i < hi; ++i) {
            list.clear();
            for (int p = aptr[i]; p < aptr[i + 1]; ++p) {
                const int k = acol[p];
                const VA a = aval[p];
                for (int q = bptr[k]; q < bptr[k + 1]; ++q) {
                    const int c = bcol[q];
                    if (marker[c] != (int)i) {
                        marker[c] = (int)i;
                        acc[c] = VC(0);
                        list.push_back(c);
                    }
                    acc[c] += VC(a) * VC(bval[q]);
                }
            }
            std::sort(list.begin(), list.end());
            for (int c : list) {
                oc.push_back(c);
                ov.push_back(acc[c]);
            }
            cptr[i + 1] = (int)list.size();
        }
    });
    stitch(n, nparts, cptr, pcol, pval, ccol, cval);
}

// C = R (A P) row by row of C, without the intermediate A P (20 M entries per plane at 1M unknowns: five such products at once
// spent their time in the allocator and in page faults, and no number of threads helped).  For coarse row I the rows i of R are
// walked in order; row i of A P is formed in a scratch accumulator exactly as spgemm would form it and added with the weight
// R[I,i]: every entry of C sees the same operations in the same order as in the two-step product -- the result is the same bit
// for bit -- at ~3.5 x its multiplications (a row of A P is formed once per coarse row that uses it), which threads now divide.
template <class VA, class VC>
static void galerkin_rowwise(const CsrD &R, int64_t an, const std::vector<int> &aptr, const std::vector<int> &acol, const std::vector<VA> &aval,
                             const CsrD &P, std::vector<int> &cptr, std::vector<int> &ccol, std::vector<VC> &cval, int nthreads) {
    (void)an;
    const int64_t n = R.n, m = P.m;
    cptr.assign(n + 1, 0);
    const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, n / 1024 + 1));
    std::vector<std::vector<int>> pcol(nparts);
    std::vector<std::vector<VC>> pval(nparts);
    auto body = [&](int64_t lo, int64_t hi, int part) {
        std::vector<int> markC(m, -1), markT(m, -1), listC, listT;
        std::vector<VC> accC(m), accT(m);
        int stamp = 0;
        std::vector<int> &oc = pcol[part];
        std::vector<VC> &ov = pval[part];
        {   // (estimate of this part's output: (R A P) has about as many entries as the rows of A that its R rows gather)
            size_t est = 0;
            for (int p = R.ptr[lo]; p < R.ptr[hi]; ++p) est += (size_t)(aptr[R.col[p] + 1] - aptr[R.col[p]]);
            est = est / 6 + 1024;
            huge_reserve(oc, est); huge_reserve(ov, est);
        }
        for (int64_t I = lo; I < hi; ++I) {
            listC.clear();
            for (int p = R.ptr[I]; p < R.ptr[I + 1]; ++p) {
                const int i = R.col[p];
                const double r = R.val[p];
                listT.clear();
                ++stamp;
                for (int q = aptr[i]; q < aptr[i + 1]; ++q) {
                    const int k = acol[q];
                    const VA a = aval[q];
                    for (int t = P.ptr[k]; t < P.ptr[k + 1]; ++t) {
                        const int J = P.col[t];
                        if (markT[J] != stamp) { markT[J] = stamp; accT[J] = VC(0); listT.push_back(J); }
                        accT[J] += VC(a) * VC(P.val[t]);
                    }
                }
                for (int J : listT) {
                    if (markC[J] != (int)I) { markC[J] = (int)I; accC[J] = VC(0); listC.push_back(J); }
                    accC[J] += VC(r) * VC(accT[J]);
                }
            }
            std::sort(listC.begin(), listC.end());
            for (int J : listC) { oc.push_back(J); ov.push_back(accC[J]); }
            cptr[I + 1] = (int)listC.size();
        }
    };
    if (nparts == 1) body(0, n, 0);
    else {
        std::vector<std::future<void>> jobs;
        for (int t = 0; t < nparts; ++t) {
            const int64_t lo = n * t / nparts, hi = n * (t + 1) / nparts;
            jobs.push_back(std::async(std::launch::async, [&body, lo, hi, t]() { body(lo, hi, t); }));
        }
        for (auto &j : jobs) j.get();
    }
    stitch(n, nparts, cptr, pcol, pval, ccol, cval);
}
// The same for NV matrices that share one pattern (K and M of a Helmholtz family: the two large planes): the index traversal -- what
// the product costs -- is done once, every entry carries NV values.  Same operations on every value as galerkin_rowwise.
template <int NV>
static void galerkin_rowwise_multi(const CsrD &R, const std::vector<int> &aptr, const std::vector<int> &acol, const std::vector<zc> *const aval[NV],
                                   const CsrD &P, std::vector<int> &cptr, std::vector<int> &ccol, std::vector<zc> *const cval[NV], int nthreads) {
    struct Vals { zc v[NV]; };
    const int64_t n = R.n, m = P.m;
    cptr.assign(n + 1, 0);
    const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, n / 1024 + 1));
    std::vector<std::vector<int>> pcol(nparts);
    std::vector<std::vector<Vals>> pval(nparts);
    auto body = [&](int64_t lo, int64_t hi, int part) {
        std::vector<int> markC(m, -1), markT(m, -1), listC, listT;
        std::vector<Vals> accC(m), accT(m);
        int stamp = 0;
        std::vector<int> &oc = pcol[part];
        std::vector<Vals> &ov = pval[part];
        {
            size_t est = 0;
            for (int p = R.ptr[lo]; p < R.ptr[hi]; ++p) est += (size_t)(aptr[R.col[p] + 1] - aptr[R.col[p]]);
            est = est / 6 + 1024;
            huge_reserve(oc, est); huge_reserve(ov, est);
        }
        for (int64_t I = lo; I < hi; ++I) {
            listC.clear();
            for (int p = R.ptr[I]; p < R.ptr[I + 1]; ++p) {
                const int i = R.col[p];
                const double r = R.val[p];
                listT.clear();
                ++stamp;
                for (int q = aptr[i]; q < aptr[i + 1]; ++q) {
                    const int k = acol[q];
                    zc a[NV];
                    for (int u = 0; u < NV; ++u) a[u] = (*aval[u])[q];
                    for (int t = P.ptr[k]; t < P.ptr[k + 1]; ++t) {
                        const int J = P.col[t];
                        if (markT[J] != stamp) { markT[J] = stamp; for (int u = 0; u < NV; ++u) accT[J].v[u] = zc(0); listT.push_back(J); }
                        for (int u = 0; u < NV; ++u) accT[J].v[u] += a[u] * zc(P.val[t]);
                    }
                }
                for (int J : listT) {
                    if (markC[J] != (int)I) { markC[J] = (int)I; for (int u = 0; u < NV; ++u) accC[J].v[u] = zc(0); listC.push_back(J); }
                    for (int u = 0; u < NV; ++u) accC[J].v[u] += zc(r) * accT[J].v[u];
                }
            }
            std::sort(listC.begin(), listC.end());
            for (int J : listC) { oc.push_back(J); ov.push_back(accC[J]); }
            cptr[I + 1] = (int)listC.size();
        }
    };
    if (nparts == 1) body(0, n, 0);
    else {
        std::vector<std::future<void>> jobs;
        for (int t = 0; t < nparts; ++t) {
            const int64_t lo = n * t / nparts, hi = n * (t + 1) / nparts;
            jobs.push_back(std::async(std::launch::async, [&body, lo, hi, t]() { body(lo, hi, t); }));
        }
        for (auto &j : jobs) j.get();
    }
    // row pointers, then the columns once and the NV value arrays
    for (int64_t i = 0; i < n; ++i) cptr[i + 1] += cptr[i];
    const size_t total = (size_t)cptr[n];
    huge_reserve(ccol, total);
    ccol.resize(total);
    for (int u = 0; u < NV; ++u) { huge_reserve(*cval[u], total); cval[u]->resize(total); }
    {
        std::vector<std::future<void>> jobs;
        size_t off = 0;
        for (int t = 0; t < nparts; ++t) {
            const size_t cnt = pcol[t].size();
            jobs.push_back(std::async(std::launch::async, [&, t, off, cnt]() {
                std::copy(pcol[t].begin(), pcol[t].end(), ccol.begin() + off);
                for (size_t e = 0; e < cnt; ++e)
                    for (int u = 0; u < NV; ++u) (*cval[u])[off + e] = pval[t][e].v[u];
            }));
            off += cnt;
        }
        for (auto &j : jobs) j.get();
    }
}
// two planes of one pattern at once (bit-identical to two galerkin() calls)
void galerkin_pair(const CsrD &R, const CsrZ &A0, const CsrZ &A1, const CsrD &P, CsrZ &C0, CsrZ &C1, int nthreads) {
    C0.n = C1.n = R.n; C0.m = C1.m = P.m;
    const std::vector<zc> *const av[2] = {&A0.val, &A1.val};
    std::vector<zc> *const cv[2] = {&C0.val, &C1.val};
    galerkin_rowwise_multi<2>(R, A0.ptr, A0.col, av, P, C0.ptr, C0.col, cv, nthreads);
    C1.ptr = C0.ptr; C1.col = C0.col;
}
// (worth it for short rows: fine level, 15 entries per row: 0.99 -> 0.45 s for the five products at 1M unknowns, 16 threads; the
// first coarse level with 48 per row: 0.07 -> 0.16 s)
static bool rowwise_on(int64_t rows, size_t nnz) {
    static const bool on = !(getenv("WAE_GALERKIN_ROWWISE") && atoi(getenv("WAE_GALERKIN_ROWWISE")) == 0);
    return on && nnz <= (size_t)24 * (size_t)std::max<int64_t>(rows, 1);
}

CsrZ galerkin(const CsrD &R, const CsrZ &A, const CsrD &P, int nthreads) {
    if (rowwise_on(A.n, A.col.size())) {
        CsrZ C;
        C.n = R.n; C.m = P.m;
        galerkin_rowwise<zc, zc>(R, A.n, A.ptr, A.col, A.val, P, C.ptr, C.col, C.val, nthreads);
        return C;
    }
    CsrZ T;
    T.n = A.n; T.m = P.m;
    spgemm<zc, double, zc>(A.n, P.m, A.ptr, A.col, A.val, P.ptr, P.col, P.val, T.ptr, T.col, T.val, nthreads);
    CsrZ C;
    C.n = R.n; C.m = P.m;
    spgemm<double, zc, zc>(R.n, P.m, R.ptr, R.col, R.val, T.ptr, T.col, T.val, C.ptr, C.col, C.val, nthreads);
    return C;
}
CsrZ galerkin(const CsrD &R, const CsrZ &A, const CsrD &P) { return galerkin(R, A, P, 1); }
CsrD galerkin_real(const CsrD &R, const CsrD &A, const CsrD &P, int nthreads) {
    if (rowwise_on(A.n, A.col.size())) {
        CsrD C;
        C.n = R.n; C.m = P.m;
        galerkin_rowwise<double, double>(R, A.n, A.ptr, A.col, A.val, P, C.ptr, C.col, C.val, nthreads);
        return C;
    }
    CsrD T;
    T.n = A.n; T.m = P.m;
    spgemm<double, double, double>(A.n, P.m, A.ptr, A.col, A.val, P.ptr, P.col, P.val, T.ptr, T.col, T.val, nthreads);
    CsrD C;
    C.n = R.n; C.m = P.m;
    spgemm<double, double, double>(R.n, P.m, R.ptr, R.col, R.val, T.ptr, T.col, T.val, C.ptr, C.col, C.val, nthreads);
    return C;
}
CsrD galerkin_real(const CsrD &R, const CsrD &A, const CsrD &P) { return galerkin_real(R, A, P, 1); }

CsrZ csr_lincomb(const std::vector<CsrZ> &planes, const std::vector<zc> &coef) {
    CsrZ C;
    if (planes.empty()) return C;
    const int64_t n = planes[0].n, m = planes[0].m;
    C.n = n; C.m = m;
    C.ptr.assign(n + 1, 0);
    const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(setup_threads(), n / 4096 + 1));
    std::vector<std::vector<int>> pcol(nparts);
    std::vector<std::vector<zc>> pval(nparts);
    (void)m;
    parallel_ranges(n, nparts, [&](int64_t lo, int64_t hi, int part) {
        // a row's entries of all planes in one short list, ordered by column with the plane order kept among equal columns: the
        // sums are formed in the order of the dense-accumulator version (acc = 0; acc += c_k a_k, k ascending) -- same bits -- without
        // its two m-sized arrays per thread
        std::vector<std::pair<int64_t, zc>> row;             // key = column * planes + plane (unique: an in-place sort keeps the plane order)
        const int64_t np = (int64_t)planes.size();
        {   // (an upper bound of this part's output: no regrowth -- every doubling of a 100-MB vector is a copy and a set of fresh pages)
            size_t bound = 0;
            for (const CsrZ &A : planes) bound += (size_t)(A.ptr[hi] - A.ptr[lo]);
            huge_reserve(pcol[part], bound); huge_reserve(pval[part], bound);
        }
        for (int64_t i = lo; i < hi; ++i) {
            row.clear();
            for (size_t k = 0; k < planes.size(); ++k) {
                const CsrZ &A = planes[k];
                for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) row.emplace_back((int64_t)A.col[p] * np + (int64_t)k, coef[k] * A.val[p]);
            }
            std::sort(row.begin(), row.end(), [](const std::pair<int64_t, zc> &x, const std::pair<int64_t, zc> &y) { return x.first < y.first; });
            for (auto &e : row) e.first /= np;
            int cnt = 0;
            for (size_t e = 0; e < row.size();) {
                zc acc = 0;
                size_t f = e;
                for (; f < row.size() && row[f].first == row[e].first; ++f) acc += row[f].second;
                pcol[part].push_back((int)row[e].first); pval[part].push_back(acc);
                ++cnt;
                e = f;
            }
            C.ptr[i + 1] = cnt;
        }
    });
    stitch(n, nparts, C.ptr, pcol, pval, C.col, C.val);
    return C;
}

// ----------------------------------------------------------------------------------------------------
// smoothed aggregation
// ----------------------------------------------------------------------------------------------------
static double diag_of(const CsrD &S, int64_t i) {
    for (int p = S.ptr[i]; p < S.ptr[i + 1]; ++p)
        if (S.col[p] == i) return S.val[p];
    return 0.0;
}

// Returns P (n x nc).  `skip[i]` != 0 marks penalty (Dirichlet-like) rows: they are neither aggregated nor
// interpolated (empty P row), so coarse spaces satisfy the essential condition exactly.
CsrD build_prolongator(const CsrD &S, const std::vector<char> &skip, double theta, bool smooth, const std::vector<int> *visit) {
    const int64_t n = S.n;
    // visit (optional): the order in which the greedy passes take the nodes (default: index order).  The library renumbers the
    // fine rows into tiles (tiles.h); visiting them in the CALLER's order keeps the aggregates -- and with them the whole
    // hierarchy and the iteration counts -- what they were before the renumbering.
    auto node = [&](int64_t k) -> int64_t { return visit ? (int64_t)(*visit)[k] : k; };
    const bool pdbg = getenv("WAE_SETUP_DEBUG") != nullptr && n > 100000;
    double tp0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    auto plap = [&](const char *what) {
        if (!pdbg) return;
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        fprintf(stderr, "[amg]   (prolongator) %-16s %.3f s\n", what, t - tp0);
        tp0 = t;
    };
    std::vector<double> D(n);
    parallel_ranges(n, setup_threads(), [&](int64_t lo, int64_t hi, int) { for (int64_t i = lo; i < hi; ++i) D[i] = std::fabs(diag_of(S, i)); });
    // strength graph (symmetric criterion), restricted to non-skipped nodes
    std::vector<int> gptr(n + 1, 0), gcol;
    {
        const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(setup_threads(), n / 4096 + 1));
        std::vector<std::vector<int>> pcol(nparts);
        parallel_ranges(n, nparts, [&](int64_t lo, int64_t hi, int part) {
            std::vector<int> &oc = pcol[part];
            huge_reserve(oc, (size_t)(S.ptr[hi] - S.ptr[lo]));
            for (int64_t i = lo; i < hi; ++i) {
                int cnt = 0;
                if (!skip[i])
                    for (int p = S.ptr[i]; p < S.ptr[i + 1]; ++p) {
                        const int j = S.col[p];
                        if (j == i || skip[j]) continue;
                        if (std::fabs(S.val[p]) >= theta * std::sqrt(D[i] * D[j])) { oc.push_back(j); ++cnt; }
                    }
                gptr[i + 1] = cnt;
            }
        });
        for (int64_t i = 0; i < n; ++i) gptr[i + 1] += gptr[i];
        gcol.resize((size_t)gptr[n]);
        size_t off = 0;
        for (auto &v : pcol) { std::copy(v.begin(), v.end(), gcol.begin() + off); off += v.size(); }
    }
    plap("strength graph");
    std::vector<int> agg(n, -1);
    int na = 0;
    for (int64_t k = 0; k < n; ++k) {   // pass 1: root nodes whose whole strong neighbourhood is free
        const int64_t i = node(k);
        if (skip[i] || agg[i] >= 0 || gptr[i + 1] == gptr[i]) continue;
        bool free_nb = true;
        for (int p = gptr[i]; p < gptr[i + 1]; ++p)
            if (agg[gcol[p]] >= 0) { free_nb = false; break; }
        if (!free_nb) continue;
        agg[i] = na;
        for (int p = gptr[i]; p < gptr[i + 1]; ++p) agg[gcol[p]] = na;
        ++na;
    }
    std::vector<int> agg2(agg);
    for (int64_t k = 0; k < n; ++k) {   // pass 2: join a neighbouring aggregate
        const int64_t i = node(k);
        if (skip[i] || agg[i] >= 0) continue;
        for (int p = gptr[i]; p < gptr[i + 1]; ++p)
            if (agg[gcol[p]] >= 0) { agg2[i] = agg[gcol[p]]; break; }
    }
    agg.swap(agg2);
    for (int64_t k = 0; k < n; ++k) {   // pass 3: leftovers form their own aggregates
        const int64_t i = node(k);
        if (skip[i] || agg[i] >= 0) continue;
        agg[i] = na;
        for (int p = gptr[i]; p < gptr[i + 1]; ++p)
            if (agg[gcol[p]] < 0) agg[gcol[p]] = na;
        ++na;
    }
    // (experiment, WAE_AMG_MERGE=1: the aggregates are merged in strongly connected pairs -- about twice as large, half as many
    // coarse unknowns; greedy matching in aggregate order, deterministic)
    static const int merge_env = getenv("WAE_AMG_MERGE") ? atoi(getenv("WAE_AMG_MERGE")) : 0;
    if (merge_env && visit != nullptr && na > 1) {                 // (fine level only: `visit` is passed for it)
        std::vector<int> mate(na, -1);
        std::vector<std::vector<int>> nbrs(na);
        for (int64_t i = 0; i < n; ++i) {
            const int a = agg[i];
            if (a < 0) continue;
            for (int p = gptr[i]; p < gptr[i + 1]; ++p) {
                const int b = agg[gcol[p]];
                if (b >= 0 && b != a) nbrs[a].push_back(b);
            }
        }
        for (int a = 0; a < na; ++a) {
            if (mate[a] >= 0) continue;
            // the free neighbour with the most strong couplings
            std::sort(nbrs[a].begin(), nbrs[a].end());
            int best = -1, bestc = 0;
            for (size_t k = 0; k < nbrs[a].size();) {
                size_t e = k;
                while (e < nbrs[a].size() && nbrs[a][e] == nbrs[a][k]) ++e;
                const int b = nbrs[a][k], c = (int)(e - k);
                if (mate[b] < 0 && b != a && c > bestc) { best = b; bestc = c; }
                k = e;
            }
            if (best >= 0) { mate[a] = best; mate[best] = a; }
            else mate[a] = a;
        }
        std::vector<int> relabel(na, -1);
        int nn = 0;
        for (int a = 0; a < na; ++a) {
            if (relabel[a] >= 0) continue;
            relabel[a] = nn;
            if (mate[a] != a && mate[a] >= 0) relabel[mate[a]] = nn;
            ++nn;
        }
        for (int64_t i = 0; i < n; ++i) if (agg[i] >= 0) agg[i] = relabel[agg[i]];
        na = nn;
    }
    plap("aggregation");
    // tentative prolongator (piecewise constant)
    CsrD Pt;
    Pt.n = n; Pt.m = na;
    Pt.ptr.assign(n + 1, 0);
    for (int64_t i = 0; i < n; ++i) {
        if (agg[i] >= 0) { Pt.col.push_back(agg[i]); Pt.val.push_back(1.0); }
        Pt.ptr[i + 1] = (int)Pt.col.size();
    }
    if (!smooth) return Pt;
    // filtered matrix F: strong couplings + diagonal, weak couplings lumped onto the diagonal (rows independent: host threads)
    CsrD F;
    F.n = n; F.m = n;
    F.ptr.assign(n + 1, 0);
    std::vector<double> Fd(n, 1.0);
    {
        const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(setup_threads(), n / 4096 + 1));
        std::vector<std::vector<int>> pcol(nparts);
        std::vector<std::vector<double>> pval(nparts);
        parallel_ranges(n, nparts, [&](int64_t lo, int64_t hi, int part) {
            std::vector<int> &oc = pcol[part];
            std::vector<double> &ov = pval[part];
            huge_reserve(oc, (size_t)(S.ptr[hi] - S.ptr[lo]) + (size_t)(hi - lo)); huge_reserve(ov, (size_t)(S.ptr[hi] - S.ptr[lo]) + (size_t)(hi - lo));
            for (int64_t i = lo; i < hi; ++i) {
                const size_t row0 = oc.size();
                if (!skip[i]) {
                    double dg = 0.0;
                    int gp = gptr[i];
                    size_t dpos = (size_t)-1;
                    for (int p = S.ptr[i]; p < S.ptr[i + 1]; ++p) {
                        const int j = S.col[p];
                        if (j == i) { dg += S.val[p]; dpos = oc.size(); oc.push_back(j); ov.push_back(0.0); continue; }
                        if (skip[j]) continue;           // coupling to a penalty node: eliminated (value there is ~0)
                        // strong neighbours are listed in column order in gcol
                        while (gp < gptr[i + 1] && gcol[gp] < j) ++gp;
                        if (gp < gptr[i + 1] && gcol[gp] == j) { oc.push_back(j); ov.push_back(S.val[p]); }
                        else dg += S.val[p];
                    }
                    if (dpos == (size_t)-1) {   // structurally missing diagonal: insert it in column order
                        oc.push_back((int)i); ov.push_back(0.0);
                        size_t q = oc.size() - 1;
                        while (q > row0 && oc[q - 1] > oc[q]) { std::swap(oc[q - 1], oc[q]); std::swap(ov[q - 1], ov[q]); --q; }
                        dpos = q;
                    }
                    if (dg == 0.0) dg = 1.0;
                    ov[dpos] = dg;
                    Fd[i] = dg;
                }
                F.ptr[i + 1] = (int)(oc.size() - row0);
            }
        });
        stitch(n, nparts, F.ptr, pcol, pval, F.col, F.val);
    }
    plap("filtered matrix");
    // two-step form (test hook g_amg_fused_prolongator = 0): F * P_tentative as a sparse product of its own, beside the power iteration
    CsrD FP;
    FP.n = n; FP.m = na;
    const bool fused = g_amg_fused_prolongator != 0;
    const int th_half = fused ? setup_threads() : std::max(1, setup_threads() / 2);
    std::future<void> fp_job;
    if (!fused)
        fp_job = std::async(std::launch::async, [&]() {
            spgemm<double, double, double>(n, na, F.ptr, F.col, F.val, Pt.ptr, Pt.col, Pt.val, FP.ptr, FP.col, FP.val, th_half);
        });
    // spectral radius of D^-1 F by power iteration
    std::vector<double> x(n), y(n);
    uint64_t lcg = 88172645463325252ull;
    for (int64_t i = 0; i < n; ++i) { lcg ^= lcg << 13; lcg ^= lcg >> 7; lcg ^= lcg << 17; x[i] = (double)(lcg % 2000) / 1000.0 - 1.0; }
    double rho = 1.0;
    for (int it = 0; it < 20; ++it) {
        parallel_ranges(n, th_half, [&](int64_t lo, int64_t hi, int) {
            for (int64_t i = lo; i < hi; ++i) {
                double s = 0.0;
                for (int p = F.ptr[i]; p < F.ptr[i + 1]; ++p) s += F.val[p] * x[F.col[p]];
                y[i] = s / Fd[i];
            }
        });
        double nrm = 0.0;                                    // (serial sum: the result does not depend on the thread count)
        for (int64_t i = 0; i < n; ++i) nrm += y[i] * y[i];
        nrm = std::sqrt(nrm);
        if (nrm == 0.0) break;
        rho = nrm;
        parallel_ranges(n, th_half, [&](int64_t lo, int64_t hi, int) { for (int64_t i = lo; i < hi; ++i) x[i] = y[i] / nrm; });
    }
    // note: with x normalised each sweep, ||D^-1 F x|| -> rho
    const double omega = (4.0 / 3.0) / rho;
    // P = Pt - omega * D^-1 F Pt
    plap("spectral radius");
    CsrD P;
    P.n = n; P.m = na;
    P.ptr.assign(n + 1, 0);
    if (fused) {
        // One pass over the rows of F: P_tentative has one unit entry per row, so row i of F * P_tentative is the row of F summed by
        // aggregate -- in the column order of F, as the sparse product would sum it (same bits) --, and the row of P follows at once.
        const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(setup_threads(), n / 4096 + 1));
        std::vector<std::vector<int>> pcol(nparts);
        std::vector<std::vector<double>> pval(nparts);
        parallel_ranges(n, nparts, [&](int64_t lo, int64_t hi, int part) {
            std::vector<int> &oc = pcol[part];
            std::vector<double> &ov = pval[part];
            huge_reserve(oc, (size_t)(F.ptr[hi] - F.ptr[lo]) + (size_t)(hi - lo)); huge_reserve(ov, (size_t)(F.ptr[hi] - F.ptr[lo]) + (size_t)(hi - lo));
            std::vector<std::pair<int, double>> row;
            for (int64_t i = lo; i < hi; ++i) {
                const size_t row0 = oc.size();
                if (!skip[i]) {
                    row.clear();
                    for (int p = F.ptr[i]; p < F.ptr[i + 1]; ++p) {
                        const int c = agg[F.col[p]];
                        if (c < 0) continue;                         // (an empty row of P_tentative)
                        size_t k = 0;
                        while (k < row.size() && row[k].first != c) ++k;
                        if (k == row.size()) row.emplace_back(c, 0.0);
                        row[k].second += F.val[p] * 1.0;
                    }
                    std::sort(row.begin(), row.end(), [](const std::pair<int, double> &x, const std::pair<int, double> &y) { return x.first < y.first; });
                    const int a = agg[i];
                    bool seen = false;
                    for (const auto &e : row) {
                        double v = -omega * e.second / Fd[i];
                        if (e.first == a) { v += 1.0; seen = true; }
                        oc.push_back(e.first);
                        ov.push_back(v);
                    }
                    if (!seen && a >= 0) {   // keep sorted order
                        oc.push_back(a);
                        ov.push_back(1.0);
                        size_t q = oc.size() - 1;
                        while (q > row0 && oc[q - 1] > oc[q]) { std::swap(oc[q - 1], oc[q]); std::swap(ov[q - 1], ov[q]); --q; }
                    }
                }
                P.ptr[i + 1] = (int)(oc.size() - row0);
            }
        });
        stitch(n, nparts, P.ptr, pcol, pval, P.col, P.val);
    } else {
        fp_job.get();
        const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(setup_threads(), n / 4096 + 1));
        std::vector<std::vector<int>> pcol(nparts);
        std::vector<std::vector<double>> pval(nparts);
        parallel_ranges(n, nparts, [&](int64_t lo, int64_t hi, int part) {
            std::vector<int> &oc = pcol[part];
            std::vector<double> &ov = pval[part];
            huge_reserve(oc, (size_t)(FP.ptr[hi] - FP.ptr[lo]) + (size_t)(hi - lo)); huge_reserve(ov, (size_t)(FP.ptr[hi] - FP.ptr[lo]) + (size_t)(hi - lo));
            for (int64_t i = lo; i < hi; ++i) {
                const size_t row0 = oc.size();
                if (!skip[i]) {
                    const int a = agg[i];
                    bool seen = false;
                    for (int p = FP.ptr[i]; p < FP.ptr[i + 1]; ++p) {
                        double v = -omega * FP.val[p] / Fd[i];
                        if (FP.col[p] == a) { v += 1.0; seen = true; }
                        oc.push_back(FP.col[p]);
                        ov.push_back(v);
                    }
                    if (!seen && a >= 0) {   // keep sorted order
                        oc.push_back(a);
                        ov.push_back(1.0);
                        size_t q = oc.size() - 1;
                        while (q > row0 && oc[q - 1] > oc[q]) { std::swap(oc[q - 1], oc[q]); std::swap(ov[q - 1], ov[q]); --q; }
                    }
                }
                P.ptr[i + 1] = (int)(oc.size() - row0);
            }
        });
        stitch(n, nparts, P.ptr, pcol, pval, P.col, P.val);
    }
    plap("assembly");
    return P;
}

void amg_setup(const std::vector<CsrZ> &planes, const std::vector<zc> &pc_ref, const AmgOptions &opt, std::vector<AmgLevel> &levels,
               std::vector<char> *penalty_rows, const std::vector<zc> *pc_shape, const std::vector<int> *visit0,
               const std::function<void(const AmgLevel &)> &on_level) {
    levels.clear();
    levels.reserve((size_t)opt.max_levels + 1);             // (on_level may keep a reference to a level while the next ones are built)
    const bool dbg = getenv("WAE_SETUP_DEBUG") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tq = now();
    auto lap = [&](const char *what) { if (dbg) { const double t = now(); fprintf(stderr, "[amg] %-28s %.3f s\n", what, t - tq); tq = t; } };
    // Real shape matrix S for strength-of-connection / aggregation / prolongator smoothing: the real part of the
    // STRUCTURALLY SYMMETRIC part of A_ref.  One-sided couplings (the flame term Q = s g^T couples every flame
    // node to the few reference nodes, Helmholtz.jl:464-487) are long-range and non-elliptic: letting them into
    // the strength graph doubles the GMRES iteration count on the 200k-DoF annulus (dev/gpu_solve_check.py).
    const int64_t n0 = planes.empty() ? 0 : planes[0].n;
    CsrD S;
    S.n = S.m = n0;
    S.ptr.assign(n0 + 1, 0);
    std::vector<double> dabs(n0);
    if (!pc_shape) {
        // One pass, without the linear combination A_ref = sum_k c_k A_k as a matrix of its own (290 MB written and read again at
        // 1M unknowns): a row's entries of all planes are merged in a short list (columns ascending, plane order kept among equal
        // columns: the sums have the bits of csr_lincomb), the mirror entry a_ji is summed from the planes' rows j (sorted columns:
        // binary search), rows are independent.  Same S, same |diagonal| as the two-step form below.
        const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(setup_threads(), n0 / 4096 + 1));
        std::vector<std::vector<int>> pcol(nparts);
        std::vector<std::vector<double>> pval(nparts);
        parallel_ranges(n0, nparts, [&](int64_t lo, int64_t hi, int part) {
            std::vector<int> &oc = pcol[part];
            std::vector<double> &ov = pval[part];
            {
                size_t bound = 0;
                for (const CsrZ &A : planes) bound += (size_t)(A.ptr[hi] - A.ptr[lo]);
                huge_reserve(oc, bound); huge_reserve(ov, bound);
            }
            // (the planes' rows are sorted: a k-way merge walks them together -- smallest head column first, the planes that have it
            // summed in plane order, as csr_lincomb sums them)
            const size_t np = planes.size();
            std::vector<int> head(np), tail(np);
            for (int64_t i = lo; i < hi; ++i) {
                for (size_t k = 0; k < np; ++k) { head[k] = planes[k].ptr[i]; tail[k] = planes[k].ptr[i + 1]; }
                int cnt = 0;
                zc dg = 0;
                for (;;) {
                    int j = INT32_MAX;
                    for (size_t k = 0; k < np; ++k)
                        if (head[k] < tail[k]) j = std::min(j, planes[k].col[head[k]]);
                    if (j == INT32_MAX) break;
                    zc aij = 0;
                    for (size_t k = 0; k < np; ++k)
                        if (head[k] < tail[k] && planes[k].col[head[k]] == j) { aij += pc_ref[k] * planes[k].val[head[k]]; ++head[k]; }
                    bool keep = j == i;
                    if (keep) dg = aij;
                    else {
                        zc aji = 0;
                        bool found = false;
                        for (size_t k = 0; k < np; ++k) {
                            const CsrZ &A = planes[k];
                            const int *b = A.col.data() + A.ptr[j], *en = A.col.data() + A.ptr[j + 1];
                            if (b == en) continue;
                            const int *q = std::lower_bound(b, en, (int)i);
                            if (q != en && *q == (int)i) { aji += pc_ref[k] * A.val[(size_t)(q - A.col.data())]; found = true; }
                        }
                        if (found) {
                            const double x = std::abs(aij), y = std::abs(aji);
                            keep = y >= 0.25 * x && x >= 0.25 * y;
                        }
                    }
                    if (keep) { oc.push_back(j); ov.push_back(aij.real()); ++cnt; }
                }
                S.ptr[i + 1] = cnt;
                dabs[i] = std::abs(dg);
            }
        });
        stitch(n0, nparts, S.ptr, pcol, pval, S.col, S.val);
        lap("shape matrix (fused)");
    } else {
        const CsrZ Afull = csr_lincomb(planes, pc_ref);
        lap("lincomb");
        const CsrZ Ashape = pc_shape ? csr_lincomb(planes, *pc_shape) : CsrZ();
        const CsrZ &Aref = pc_shape ? Ashape : Afull;          // the shape matrix comes from here, the penalty test from Afull
        {
            // (the mirror entry a_ji is looked up in row j -- columns are sorted -- instead of in a transposed copy: rows are
            // independent, so the host threads divide them; same entries, same values as the transpose-based loop)
            const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(setup_threads(), n0 / 4096 + 1));
            std::vector<std::vector<int>> pcol(nparts);
            std::vector<std::vector<double>> pval(nparts);
            parallel_ranges(n0, nparts, [&](int64_t lo, int64_t hi, int part) {
                std::vector<int> &oc = pcol[part];
                std::vector<double> &ov = pval[part];
                huge_reserve(oc, (size_t)(Aref.ptr[hi] - Aref.ptr[lo])); huge_reserve(ov, (size_t)(Aref.ptr[hi] - Aref.ptr[lo]));
                for (int64_t i = lo; i < hi; ++i) {
                    int cnt = 0;
                    for (int p = Aref.ptr[i]; p < Aref.ptr[i + 1]; ++p) {
                        const int j = Aref.col[p];
                        bool keep = j == i;
                        if (!keep) {
                            const int *b = Aref.col.data() + Aref.ptr[j], *e = Aref.col.data() + Aref.ptr[j + 1];
                            const int *f = std::lower_bound(b, e, (int)i);
                            if (f != e && *f == (int)i) {
                                const double aij = std::abs(Aref.val[p]);
                                const double aji = std::abs(Aref.val[(size_t)(f - Aref.col.data())]);
                                keep = aji >= 0.25 * aij && aij >= 0.25 * aji;
                            }
                        }
                        if (keep) { oc.push_back(j); ov.push_back(Aref.val[p].real()); ++cnt; }
                    }
                    S.ptr[i + 1] = cnt;
                }
            });
            stitch(n0, nparts, S.ptr, pcol, pval, S.col, S.val);
        }
        for (int64_t i = 0; i < n0; ++i) {
            zc dg = 0;
            for (int p = Afull.ptr[i]; p < Afull.ptr[i + 1]; ++p)
                if (Afull.col[p] == i) dg = Afull.val[p];
            dabs[i] = std::abs(dg);
        }
    }
    std::vector<char> skip(n0, 0);
    if (n0 > 0) {
        std::vector<double> tmp(dabs);
        std::nth_element(tmp.begin(), tmp.begin() + n0 / 2, tmp.end());
        const double med = tmp[n0 / 2];
        for (int64_t i = 0; i < n0; ++i) skip[i] = dabs[i] > opt.penalty_ratio * med;
    }
    if (penalty_rows) *penalty_rows = skip;
    lap("shape matrix + penalty rows");
    const std::vector<CsrZ> *curp = &planes;               // the planes of the level being coarsened (no copies: `levels` is reserved)
    int64_t n = n0;
    while (n > opt.max_coarse && (int)levels.size() < opt.max_levels) {
        CsrD P = build_prolongator(S, skip, opt.theta, true, levels.empty() ? visit0 : nullptr);
        lap("prolongator");
        if (P.m >= (int64_t)(0.9 * n) || P.m == 0) break;
        CsrD R = csr_transpose(P);
        AmgLevel L;
        const std::vector<CsrZ> &cur = *curp;
        // the triple products of the planes (and of the shape matrix) are independent: one host thread each
        std::vector<CsrZ> next(cur.size());
        {
            std::vector<std::future<void>> jobs;
            const int inner = std::max(1, setup_threads() / (int)(cur.size() + 1));     // threads per triple product
            // planes 0 and 1 with one pattern (K and M) and short rows: one traversal for both, with the threads of two products
            static const bool pair_on = !(getenv("WAE_GALERKIN_PAIR") && atoi(getenv("WAE_GALERKIN_PAIR")) == 0);
            const bool pair = pair_on && cur.size() >= 2 && rowwise_on(cur[0].n, cur[0].col.size()) && cur[0].ptr == cur[1].ptr && cur[0].col == cur[1].col;
            // (threads: the pair and the shape matrix are the two long jobs -- half of the threads each; the other planes -- boundary
            // and flame terms, a hundredth of the entries -- one thread each)
            const int big = pair ? std::max(1, (setup_threads() - (int)cur.size() + 2) / 2) : inner;
            if (pair) jobs.push_back(std::async(std::launch::async, [&]() { galerkin_pair(R, cur[0], cur[1], P, next[0], next[1], big); }));
            for (size_t q = pair ? 2 : 0; q < cur.size(); ++q)
                jobs.push_back(std::async(std::launch::async, [&, q]() { next[q] = galerkin(R, cur[q], P, pair ? 1 : inner); }));
            CsrD Snext = galerkin_real(R, S, P, big);
            for (auto &j : jobs) j.get();
            S = std::move(Snext);
        }
        lap("galerkin products");
        L.P = std::move(P); L.R = std::move(R);
        L.coarse_planes = std::move(next);
        levels.push_back(std::move(L));
        // (the callback -- and whatever it starts on other threads -- may change P and R of this level, not its planes: the next
        // level is built from them)
        if (on_level) on_level(levels.back());
        curp = &levels.back().coarse_planes;
        n = S.n;
        skip.assign(n, 0);
    }
}
