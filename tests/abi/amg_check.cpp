// Host-only check of csrc/amg.cpp (no GPU needed; built by tests/test_abi.py with hipcc, which provides the HIP vector types the
// internal header uses): the fused triple product of two planes of one pattern (galerkin_pair: K and M of a Helmholtz family,
// the largest item of the multigrid set-up) must equal two separate products bit for bit -- pattern and values -- for every
// thread count, with short and with long rows (the row-wise and the two-step form of galerkin()).
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
    std::printf("amg_check ok %d\n", checks);
    return 0;
}
