// Row/column reordering into compact tiles (host side) and the tile-local storage of an operator.
//
// Why: at batch width 64 one row of an interleaved multivector is 1 KB and the fused SpMV gathers ~15 of them per matrix
// row.  With the producer's numbering (lexicographic on the benchmark meshes) the rows a workgroup gathers are spread over
// three grid planes: nothing is reused inside a CU and at 1M DoF not even inside an XCD's L2 (round 1: 1.77x the algorithmic
// bytes crossed the fabric).  Here the rows are renumbered so that 256 consecutive rows form a compact brick of the mesh
// graph: their columns (the tile's "window", ~2x the rows) fit LDS, are loaded once and serve all ~15 gathers per row.
//
// The ordering needs no coordinates: three nested breadth-first stages on the pattern graph -- shells of a BFS from a
// pseudo-peripheral node, strips inside a shell, bricks inside a strip -- each a few levels thick.
#include "tiles.h"
#include "hugemem.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <future>
#include <numeric>
#include <thread>

Pattern union_pattern(const std::vector<CsrZ> &planes) {
    Pattern U;
    if (planes.empty()) return U;
    U.n = planes[0].n;
    std::vector<const CsrZ *> distinct;
    for (const CsrZ &A : planes) {
        bool seen = false;
        for (const CsrZ *B : distinct) seen = seen || csr_same_pattern(A, *B);
        if (!seen) distinct.push_back(&A);
    }
    U.ptr.assign(U.n + 1, 0);
    if (distinct.size() == 1) {
        U.ptr = distinct[0]->ptr;
        U.col = distinct[0]->col;
        return U;
    }
    std::vector<int> buf;
    for (int64_t i = 0; i < U.n; ++i) {
        buf.clear();
        for (const CsrZ *A : distinct) buf.insert(buf.end(), A->col.begin() + A->ptr[i], A->col.begin() + A->ptr[i + 1]);
        std::sort(buf.begin(), buf.end());
        buf.erase(std::unique(buf.begin(), buf.end()), buf.end());
        U.col.insert(U.col.end(), buf.begin(), buf.end());
        U.ptr[i + 1] = (int)U.col.size();
    }
    return U;
}

namespace {

struct Graph {
    int64_t n = 0;
    std::vector<int> ptr, col;
};

// symmetrised pattern without self loops; "hub" nodes (degree far above the median: the reference nodes of a flame term
// couple to every node of their flame) are cut out so that they do not short-circuit the breadth-first distances
Graph ordering_graph(const Pattern &U) {
    const int64_t n = U.n;
    std::vector<int> tptr(n + 1, 0), tcol(U.col.size());
    for (int c : U.col) tptr[c + 1]++;
    for (int64_t i = 0; i < n; ++i) tptr[i + 1] += tptr[i];
    {
        std::vector<int> pos(tptr.begin(), tptr.end() - 1);
        for (int64_t i = 0; i < n; ++i)
            for (int p = U.ptr[i]; p < U.ptr[i + 1]; ++p) tcol[pos[U.col[p]]++] = (int)i;
    }
    Graph G;
    G.n = n;
    G.ptr.assign(n + 1, 0);
    G.col.reserve(U.col.size() + U.col.size() / 8);
    for (int64_t i = 0; i < n; ++i) {                       // merge of two sorted lists
        int a = U.ptr[i], ae = U.ptr[i + 1], b = tptr[i], be = tptr[i + 1];
        while (a < ae || b < be) {
            int v;
            if (b >= be || (a < ae && U.col[a] <= tcol[b])) { v = U.col[a]; if (b < be && tcol[b] == v) ++b; ++a; }
            else { v = tcol[b]; ++b; }
            if (v != i) G.col.push_back(v);
        }
        G.ptr[i + 1] = (int)G.col.size();
    }
    std::vector<int> deg(n);
    for (int64_t i = 0; i < n; ++i) deg[i] = G.ptr[i + 1] - G.ptr[i];
    std::vector<int> tmp(deg);
    std::nth_element(tmp.begin(), tmp.begin() + n / 2, tmp.end());
    const int hub_deg = std::max(64, 8 * tmp[n / 2]);
    bool any = false;
    for (int64_t i = 0; i < n && !any; ++i) any = deg[i] > hub_deg;
    if (!any) return G;
    Graph H;
    H.n = n;
    H.ptr.assign(n + 1, 0);
    H.col.reserve(G.col.size());
    for (int64_t i = 0; i < n; ++i) {
        if (deg[i] <= hub_deg)
            for (int p = G.ptr[i]; p < G.ptr[i + 1]; ++p)
                if (deg[G.col[p]] <= hub_deg) H.col.push_back(G.col[p]);
        H.ptr[i + 1] = (int)H.col.size();
    }
    return H;
}

struct Member {            // membership of a node in the sub-graph being ordered: tags of the enclosing shell / strip
    const int *t0 = nullptr;
    int v0 = 0;
    const int *t1 = nullptr;
    int v1 = 0;
    bool operator()(int u) const { return (!t0 || t0[u] == v0) && (!t1 || t1[u] == v1); }
};

// breadth-first sweep from s over the nodes of `in` (lev = -1 on entry for all of them); q receives the visiting order
// (ascending level); returns the number of levels
int bfs(const Graph &G, const Member &in, int s, std::vector<int> &lev, std::vector<int> &q) {
    q.clear();
    q.push_back(s);
    lev[s] = 0;
    size_t head = 0;
    while (head < q.size()) {
        const int v = q[head++];
        const int lv = lev[v] + 1;
        for (int p = G.ptr[v]; p < G.ptr[v + 1]; ++p) {
            const int u = G.col[p];
            if (in(u) && lev[u] < 0) { lev[u] = lv; q.push_back(u); }    // (membership first: lev of foreign nodes may be in another thread's hands)
        }
    }
    return lev[q.back()] + 1;
}

struct Orderer {
    const Graph &G;
    int thick, leaf;
    std::vector<int> lev, tag0, tag1;
    explicit Orderer(const Graph &g, int thick_, int leaf_) : G(g), thick(thick_), leaf(leaf_), lev(g.n, -1), tag0(g.n, -1), tag1(g.n, -1) {}

    // nodes[0..cnt): a sub-graph (membership `in`), written to out[0..cnt) in the new order.  depth 0: whole graph -> shells;
    // depth 1: one shell -> strips; depth 2: one strip -> breadth-first order from one of its ends.
    void order(const int *nodes, int64_t cnt, const Member &in, int depth, int *out) {
        for (int64_t i = 0; i < cnt; ++i) lev[nodes[i]] = -1;
        std::vector<int> comp, q;
        int *o = out;
        for (int64_t i = 0; i < cnt; ++i) {
            const int v = nodes[i];
            if (lev[v] >= 0) continue;                       // belongs to a component already emitted
            bfs(G, in, v, lev, comp);
            if (comp.size() > 2) {                           // pseudo-peripheral start: far end of a sweep from the far end of the first
                for (int x : comp) lev[x] = -1;
                bfs(G, in, comp.back(), lev, q);
                const int far = q.back();
                for (int x : q) lev[x] = -1;
                bfs(G, in, far, lev, comp);
            }
            const int nl = lev[comp.back()] + 1;
            if ((int64_t)comp.size() <= leaf || depth >= 2 || nl <= thick) {
                std::copy(comp.begin(), comp.end(), o);
                o += comp.size();
                continue;
            }
            // cut the levels into slabs about `thick` levels thick; comp is in ascending level order, so a slab is a range
            const int nslab = std::max(1, (nl + thick / 2) / thick);
            std::vector<int64_t> start(nslab + 1, (int64_t)comp.size());
            {
                int k = 0;
                start[0] = 0;
                for (size_t j = 0; j < comp.size(); ++j) {
                    const int s = (int)((int64_t)lev[comp[j]] * nslab / nl);
                    while (k < s) start[++k] = (int64_t)j;
                }
                while (k < nslab) start[++k] = (int64_t)comp.size();
            }
            std::vector<int> &tg = depth == 0 ? tag0 : tag1;
            for (int s = 0; s < nslab; ++s)
                for (int64_t j = start[s]; j < start[s + 1]; ++j) tg[comp[j]] = s;
            auto run = [&](int s) {
                Member sub = in;
                if (depth == 0) { sub.t0 = tag0.data(); sub.v0 = s; }
                else { sub.t1 = tag1.data(); sub.v1 = s; }
                order(comp.data() + start[s], start[s + 1] - start[s], sub, depth + 1, o + start[s]);
            };
            if (depth == 0 && comp.size() > 100000) {
                // shells are independent: a thread per shell (disjoint nodes; a thread reads another shell's tag0 only, which is
                // constant by now, and the tag1 / lev entries of its own shell)
                const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
                std::atomic<int> next(0);
                std::vector<std::future<void>> jobs;
                for (unsigned t = 0; t < hw; ++t)
                    jobs.push_back(std::async(std::launch::async, [&]() { for (int s; (s = next.fetch_add(1)) < nslab;) run(s); }));
                for (auto &j : jobs) j.get();
            } else {
                for (int s = 0; s < nslab; ++s) run(s);
            }
            o += comp.size();
            if (depth == 1) for (int x : comp) tag1[x] = -1;       // (strip ids are local to this shell component)
        }
    }
};

}   // namespace

TilePlan plan_tiles(const Pattern &U, int tile_rows, int wcap, int thick) {
    TilePlan P;
    const int64_t n = U.n;
    if (n < 2 * (int64_t)tile_rows) return P;                 // (small dense families: nothing to gain)
    for (int64_t i = 0; i < n; ++i)
        if (U.ptr[i + 1] - U.ptr[i] > wcap) return P;        // a row wider than any window: no tiling for this operator
    std::vector<int> order0(n);
    {
        const Graph G = ordering_graph(U);
        Orderer O(G, thick, tile_rows);
        std::vector<int> all(n);
        std::iota(all.begin(), all.end(), 0);
        O.order(all.data(), n, Member(), 0, order0.data());
    }
    std::vector<int> inv0(n);
    for (int64_t i = 0; i < n; ++i) inv0[order0[i]] = (int)i;
    // cut into tiles: at most tile_rows rows, window (distinct columns) at most wcap
    std::vector<int> mark(n, -1);
    P.row_ptr.push_back(0);
    int tile = 0, rows = 0, win = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int old = order0[i];
        int fresh = 0;
        for (int p = U.ptr[old]; p < U.ptr[old + 1]; ++p) fresh += mark[U.col[p]] != tile;
        if (rows == tile_rows || win + fresh > wcap) {
            P.wmax = std::max(P.wmax, win);
            P.row_ptr.push_back((int)i);
            ++tile; rows = 0; win = 0;
            fresh = U.ptr[old + 1] - U.ptr[old];
        }
        for (int p = U.ptr[old]; p < U.ptr[old + 1]; ++p) mark[U.col[p]] = tile;
        win += fresh;
        ++rows;
    }
    P.wmax = std::max(P.wmax, win);
    P.row_ptr.push_back((int)n);
    // rows of a tile by decreasing length (stable): the 64-row slices of the tile-local storage pad to their longest row
    P.perm.resize(n);
    for (size_t t = 0; t + 1 < P.row_ptr.size(); ++t) {
        const int a = P.row_ptr[t], b = P.row_ptr[t + 1];
        std::copy(order0.begin() + a, order0.begin() + b, P.perm.begin() + a);
        std::stable_sort(P.perm.begin() + a, P.perm.begin() + b,
                         [&](int x, int y) { return U.ptr[x + 1] - U.ptr[x] > U.ptr[y + 1] - U.ptr[y]; });
    }
    P.iperm.resize(n);
    for (int64_t i = 0; i < n; ++i) P.iperm[P.perm[i]] = (int)i;
    return P;
}

// host threads of the set-up (as csrc/amg.cpp): contiguous ranges of rows / tiles, one per thread; every range writes its own part
// of the output, so the result is the serial one
static int tile_threads() {
    static const int n = []() {
        if (const char *e = getenv("WAE_SETUP_THREADS")) return std::max(1, atoi(e));
        return (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    }();
    return n;
}
template <class F> static void tile_ranges(int64_t n, int64_t grain, F &&body) {
    const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(tile_threads(), n / std::max<int64_t>(grain, 1) + 1));
    if (nparts == 1) { body((int64_t)0, n); return; }
    std::vector<std::future<void>> jobs;
    for (int t = 0; t < nparts; ++t) {
        const int64_t lo = n * t / nparts, hi = n * (t + 1) / nparts;
        jobs.push_back(std::async(std::launch::async, [&body, lo, hi]() { body(lo, hi); }));
    }
    for (auto &j : jobs) j.get();
}

CsrZ permute_symmetric(const CsrZ &A, const std::vector<int> &perm, const std::vector<int> &iperm) {
    CsrZ B;
    B.n = A.n; B.m = A.m;
    B.ptr.assign(A.n + 1, 0);
    for (int64_t i = 0; i < A.n; ++i) B.ptr[i + 1] = B.ptr[i] + (A.ptr[perm[i] + 1] - A.ptr[perm[i]]);
    huge_reserve(B.col, A.col.size()); huge_reserve(B.val, A.val.size());
    B.col.resize(A.col.size());
    B.val.resize(A.val.size());
    tile_ranges(A.n, 8192, [&](int64_t lo, int64_t hi) {
        std::vector<std::pair<int, zc>> row;
        for (int64_t i = lo; i < hi; ++i) {
            const int old = perm[i];
            row.clear();
            for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p) row.emplace_back(iperm[A.col[p]], A.val[p]);
            std::sort(row.begin(), row.end(), [](const std::pair<int, zc> &x, const std::pair<int, zc> &y) { return x.first < y.first; });
            int q = B.ptr[i];
            for (const auto &e : row) { B.col[q] = e.first; B.val[q] = e.second; ++q; }
        }
    });
    return B;
}

TileWindows build_windows(const Pattern &U, const std::vector<int> &row_ptr) {
    TileWindows W;
    const size_t nt = row_ptr.size() - 1;
    W.win_ptr.assign(nt + 1, 0);
    std::vector<std::vector<int>> per(nt);                   // sorted distinct columns of every tile, then laid end to end
    tile_ranges((int64_t)nt, 64, [&](int64_t lo, int64_t hi) {
        for (int64_t t = lo; t < hi; ++t) {
            std::vector<int> &buf = per[(size_t)t];
            buf.assign(U.col.begin() + U.ptr[row_ptr[t]], U.col.begin() + U.ptr[row_ptr[t + 1]]);
            std::sort(buf.begin(), buf.end());
            buf.erase(std::unique(buf.begin(), buf.end()), buf.end());
        }
    });
    for (size_t t = 0; t < nt; ++t) W.win_ptr[t + 1] = W.win_ptr[t] + (int)per[t].size();
    W.win_cols.resize((size_t)W.win_ptr[nt]);
    tile_ranges((int64_t)nt, 64, [&](int64_t lo, int64_t hi) {
        for (int64_t t = lo; t < hi; ++t) std::copy(per[(size_t)t].begin(), per[(size_t)t].end(), W.win_cols.begin() + W.win_ptr[(size_t)t]);
    });
    return W;
}

TileGroupHost build_tile_group(const std::vector<const CsrZ *> &mats, bool is_real, const std::vector<int> &row_ptr, const TileWindows &W, int lpr,
                               int slices) {
    const int TILE_SLICES = slices;                           // (shadows the default: 8 wavefronts per tile, or 16)
    TileGroupHost T;
    const CsrZ &A = *mats[0];
    const int np = (int)mats.size();
    const int wpe = is_real ? np : 2 * np;                    // doubles per entry
    const size_t nt = row_ptr.size() - 1;
    const int rpw = 64 / lpr;                                 // rows per wavefront
    // slice (tile, wavefront w) = rows rpw w .. rpw (w + 1) - 1 of the tile; lane lpr i + h holds the entries k = h, h + lpr, ... of row i
    T.sptr.assign(TILE_SLICES * nt + 1, 0);
    for (size_t t = 0; t < nt; ++t)
        for (int w = 0; w < TILE_SLICES; ++w) {
            int len = 0;
            for (int r = row_ptr[t] + rpw * w; r < std::min(row_ptr[t] + rpw * (w + 1), row_ptr[t + 1]); ++r) len = std::max(len, A.ptr[r + 1] - A.ptr[r]);
            T.sptr[TILE_SLICES * t + w + 1] = T.sptr[TILE_SLICES * t + w] + 64 * ((len + lpr - 1) / lpr);
        }
    const size_t total = (size_t)T.sptr.back();
    huge_reserve(T.sidx, total); huge_reserve(T.svals, total * wpe);
    T.sidx.assign(total, 0);
    T.svals.assign(total * wpe, 0.0);
    T.dslot.assign((size_t)row_ptr.back(), (unsigned short)0xFFFF);
    static const bool parity_sort = !(getenv("WAE_TILE_PARITY") && atoi(getenv("WAE_TILE_PARITY")) == 0);
    tile_ranges((int64_t)nt, 16, [&](int64_t t_lo, int64_t t_hi) {
    std::vector<std::pair<int, int>> ent, ev, od;
    for (size_t t = (size_t)t_lo; t < (size_t)t_hi; ++t) {
        const int *wc = W.win_cols.data() + W.win_ptr[t];
        const int wn = W.win_ptr[t + 1] - W.win_ptr[t];
        for (int r = row_ptr[t]; r < row_ptr[t + 1]; ++r) {
            const int lr = r - row_ptr[t], w = lr / rpw, riw = lr % rpw, lane0 = riw * lpr;
            const size_t s0 = (size_t)T.sptr[TILE_SLICES * t + w];
            // Entry order inside a row is free; it decides which window slots the lanes of an LDS cycle read together, and two
            // lanes that read the same column position share a bank group exactly when their window slots have equal parity.
            // lpr = 2: the two lanes of a row read the same position at the same time, entries 2 j and 2 j + 1: even positions
            // take the row's even-slot entries and odd positions its odd-slot entries as far as they pair up.
            // lpr = 4: the partner of lane (row i, q) is lane (row i', q) of a row whose number differs in bit 1 (kernel: rot);
            // rows with that bit clear list their even-slot entries first, the others their odd-slot entries.
            ent.clear();
            for (int p = A.ptr[r]; p < A.ptr[r + 1]; ++p) {
                ent.emplace_back((int)(std::lower_bound(wc, wc + wn, A.col[p]) - wc), p);
                if (A.col[p] == r && A.n == A.m) T.dslot[(size_t)r] = (unsigned short)ent.back().first;
            }
            if (parity_sort && lpr == 2) {
                ev.clear(); od.clear();
                for (const auto &e : ent) (e.first & 1 ? od : ev).push_back(e);
                const size_t pairs = std::min(ev.size(), od.size());
                ent.clear();
                for (size_t j = 0; j < pairs; ++j) { ent.push_back(ev[j]); ent.push_back(od[j]); }
                for (size_t j = pairs; j < ev.size(); ++j) ent.push_back(ev[j]);
                for (size_t j = pairs; j < od.size(); ++j) ent.push_back(od[j]);
            } else if (parity_sort) {
                const int first = (riw >> 1) & 1;
                std::stable_sort(ent.begin(), ent.end(), [first](const std::pair<int, int> &x, const std::pair<int, int> &y) {
                    return ((x.first & 1) ^ first) < ((y.first & 1) ^ first);
                });
            }
            for (size_t k = 0; k < ent.size(); ++k) {
                const int p = ent[k].second;
                const size_t e = s0 + (k / lpr) * 64 + lane0 + (k % lpr);
                T.sidx[e] = (unsigned short)ent[k].first;
                for (int q = 0; q < np; ++q) {
                    const zc v = mats[q]->val[p];
                    if (is_real) T.svals[e * np + q] = v.real();
                    else { T.svals[(e * np + q) * 2] = v.real(); T.svals[(e * np + q) * 2 + 1] = v.imag(); }
                }
            }
        }
    }
    });
    return T;
}
