// Host-only check of csrc/amg.cpp (no GPU needed; built by tests/test_abi.py with hipcc, which provides the HIP vector types the
// internal header uses): the fused triple product of two planes of one pattern (galerkin_pair: K and M of a Helmholtz family,
// the largest item of the multigrid set-up) must equal two separate products bit for bit -- pattern and values -- for every
// thread count, with short and with long rows (the row-wise and the two-step form of galerkin()); and the one-pass shape matrix of
// amg_setup against the two-step form (linear combination first): same hierarchy.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "amg.h"

static CsrZ random_pattern(int64_t n, int per_row, std::mt19937_64 &g) {
    CsrZ A;
    A.n = A.m = n;
    A.ptr.assign(n + 1, 0);
    std::uniform_int_distribution<int64_t> col(0, n - 1);
    std::normal_distribution<double> val;
    for (int64_t i = 0; i < n; ++i) {
        std::vector<int> c{(int)i};
        for (int k = 1; k < per_row; ++k) c.push_back((int)((i + col(g) % 97 - 48 + n) % n));
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end()), c.end());
        for (int j : c) { A.col.push_back(j); A.val.push_back(zc(val(g), val(g))); }
        A.ptr[i + 1] = (int)A.col.size();
    }
    return A;
}

static bool same(const CsrZ &X, const CsrZ &Y) {
    return X.n == Y.n && X.m == Y.m && X.ptr == Y.ptr && X.col == Y.col && X.val.size() == Y.val.size() &&
           std::memcmp(X.val.data(), Y.val.data(), X.val.size() * sizeof(zc)) == 0;
}

int main() {
    std::mt19937_64 g(12345);
    int checks = 0;
    for (int per_row : {9, 15}) {
        const int64_t n = 6000;
        CsrZ A0 = random_pattern(n, per_row, g), A1 = A0;
        std::normal_distribution<double> val;
        for (zc &v : A1.val) v = zc(val(g), 0.0);
        // prolongator: aggregates of ~7 rows, 1-3 entries per row
        CsrD P;
        P.n = n; P.m = n / 7 + 1;
        P.ptr.assign(n + 1, 0);
        std::uniform_int_distribution<int> extra(0, 2);
        for (int64_t i = 0; i < n; ++i) {
            std::vector<int> c{(int)(i / 7)};
            for (int k = extra(g); k > 0; --k) c.push_back((int)((i / 7 + k * 3) % P.m));
            std::sort(c.begin(), c.end());
            c.erase(std::unique(c.begin(), c.end()), c.end());
            for (int j : c) { P.col.push_back(j); P.val.push_back(val(g)); }
            P.ptr[i + 1] = (int)P.col.size();
        }
        const CsrD R = csr_transpose(P);
        const CsrZ C0 = galerkin(R, A0, P, 1), C1 = galerkin(R, A1, P, 1);
        for (int nt : {1, 3, 8}) {
            CsrZ D0, D1;
            galerkin_pair(R, A0, A1, P, D0, D1, nt);
            if (!same(C0, D0) || !same(C1, D1)) { std::printf("mismatch per_row=%d threads=%d\n", per_row, nt); return 1; }
            if (!same(C0, galerkin(R, A0, P, nt))) { std::printf("galerkin differs with %d threads\n", nt); return 1; }
            checks += 3;
        }
    }
    // The shape matrix of the set-up: the one-pass form (no linear combination as a matrix of its own) against the two-step form,
    // which the optional shape coefficients select -- here with the same coefficients, so both must build the same hierarchy.
    {
        const int64_t n = 20000;
        std::vector<CsrZ> planes;
        planes.push_back(random_pattern(n, 13, g));
        {   // make the first plane structurally symmetric with a dominant diagonal (an elliptic-like operator)
            CsrZ &K = planes[0];
            const CsrZ T = csr_transpose(K);
            std::vector<zc> one(2, zc(1.0));
            K = csr_lincomb({K, T}, one);
            for (int64_t i = 0; i < n; ++i)
                for (int p = K.ptr[i]; p < K.ptr[i + 1]; ++p) K.val[p] = K.col[p] == i ? zc(40.0) : zc(-0.5 - 0.4 * std::cos((double)(K.col[p] + i)));
        }
        planes.push_back(planes[0]);                         // a second plane on the same pattern
        for (zc &v : planes[1].val) v *= 0.01;
        CsrZ Q;                                              // a small one-sided term: rows 100..399 couple to columns 5, 6
        Q.n = Q.m = n;
        Q.ptr.assign(n + 1, 0);
        for (int64_t i = 0; i < n; ++i) {
            if (i >= 100 && i < 400) { Q.col.push_back(5); Q.val.push_back(zc(3.0, 1.0)); Q.col.push_back(6); Q.val.push_back(zc(-2.0, 0.5)); }
            Q.ptr[i + 1] = (int)Q.col.size();
        }
        planes.push_back(Q);
        const std::vector<zc> pc{zc(1.0), zc(-25.0, 3.0), zc(0.7, -0.2)};
        AmgOptions opt;
        std::vector<AmgLevel> la, lb;
        std::vector<char> pa, pb;
        amg_setup(planes, pc, opt, la, &pa, nullptr);
        amg_setup(planes, pc, opt, lb, &pb, &pc);
        if (la.size() != lb.size() || la.empty() || pa != pb) { std::printf("hierarchies differ in depth: %zu %zu\n", la.size(), lb.size()); return 1; }
        for (size_t l = 0; l < la.size(); ++l) {
            const bool ok = la[l].P.ptr == lb[l].P.ptr && la[l].P.col == lb[l].P.col && la[l].P.val == lb[l].P.val &&
                            la[l].coarse_planes.size() == lb[l].coarse_planes.size();
            if (!ok) { std::printf("level %zu: prolongators differ\n", l); return 1; }
            for (size_t q = 0; q < la[l].coarse_planes.size(); ++q)
                if (!same(la[l].coarse_planes[q], lb[l].coarse_planes[q])) { std::printf("level %zu plane %zu differs\n", l, q); return 1; }
            ++checks;
        }
        std::printf("hierarchy: %zu levels, first coarse size %lld\n", la.size(), (long long)la[0].P.m);
        // the prolongator smoothing in one pass over the filtered matrix against its two-step form (sparse product, then assembly)
        g_amg_fused_prolongator = 0;
        std::vector<AmgLevel> lc;
        amg_setup(planes, pc, opt, lc, nullptr, nullptr);
        g_amg_fused_prolongator = 1;
        if (lc.size() != la.size()) { std::printf("two-step prolongator: depth differs\n"); return 1; }
        for (size_t l = 0; l < la.size(); ++l) {
            if (!(la[l].P.ptr == lc[l].P.ptr && la[l].P.col == lc[l].P.col && la[l].P.val == lc[l].P.val)) { std::printf("level %zu: fused prolongator differs\n", l); return 1; }
            ++checks;
        }
    }
    std::printf("amg_check ok %d\n", checks);
    return 0;
}
