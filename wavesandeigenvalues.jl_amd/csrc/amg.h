// Host-side multigrid set-up (smoothed aggregation) -- declarations.
#pragma once
#include <functional>

#include "wae_internal.h"

struct AmgOptions {
    double theta = 0.02;          // strength-of-connection threshold
    int64_t max_coarse = 128;     // stop coarsening at or below this many unknowns
    int max_levels = 10;
    double penalty_ratio = 1e8;   // |a_ii| > ratio * median|a_jj|  =>  penalty (Dirichlet-like) row
};
struct AmgLevel {
    CsrD P, R;                            // prolongation (n_fine x n_coarse) and restriction R = P^T
    std::vector<CsrZ> coarse_planes;      // R * plane_q * P for every plane q
};
CsrZ csr_lincomb(const std::vector<CsrZ> &planes, const std::vector<zc> &coef);
CsrD galerkin_real(const CsrD &R, const CsrD &A, const CsrD &P);
CsrZ galerkin(const CsrD &R, const CsrZ &A, const CsrD &P, int nthreads);
// R A0 P and R A1 P for two matrices of one pattern in one traversal; bit-identical to two galerkin() calls (tests/abi/amg_check.cpp)
void galerkin_pair(const CsrD &R, const CsrZ &A0, const CsrZ &A1, const CsrD &P, CsrZ &C0, CsrZ &C1, int nthreads);
// test hook: 0 selects the two-step form of the prolongator smoothing (F * P_tentative as a sparse product, then the assembly); the
// default forms the rows of P in one pass over F -- the same bits (tests/abi/amg_check.cpp)
extern int g_amg_fused_prolongator;
CsrD build_prolongator(const CsrD &S, const std::vector<char> &skip, double theta, bool smooth, const std::vector<int> *visit = nullptr);
// penalty_rows (optional): receives the fine-level flags of the rows detected as penalty (Dirichlet-like) rows
// pc_shape (optional): plane coefficients of the operator the strength graph / aggregates / prolongator smoothing are taken
// from, when that should not be the full reference operator (Bloch families: without the seam couplings, so that no
// aggregate spans the seam across which the solution jumps by exp(i b 2pi/N)).
// visit0 (optional): order in which the fine-level aggregation visits the nodes (see build_prolongator).
// on_level (optional): called with every level right after its planes have been formed (the caller may start work that needs
// only that level -- the tile plan of level 1 -- on another thread while the deeper levels are built; the reference stays valid).
void amg_setup(const std::vector<CsrZ> &planes, const std::vector<zc> &pc_ref, const AmgOptions &opt, std::vector<AmgLevel> &levels,
               std::vector<char> *penalty_rows = nullptr, const std::vector<zc> *pc_shape = nullptr, const std::vector<int> *visit0 = nullptr,
               const std::function<void(const AmgLevel &)> &on_level = nullptr);
