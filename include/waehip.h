/* waehip.h -- C ABI of libwaehip.so: the MI355X (gfx950) NLEVP hot path for WavesAndEigenvalues.jl.
 *
 * The reference (Julia, serial, CPU) has no FFI of its own; the drop-in boundary is the method surface its
 * solvers use on a LinearOperatorFamily.  Each entry below names the reference call sites
 * it replaces (paths relative to the reference repository root).  Conventions:
 *   - every function returns int: 0 ok, >0 warning (e.g. WAE_WARN_MAXITER), <0 error; no C++ exception
 *     crosses the boundary; wae_last_error() returns a thread-local message for the last failure.
 *   - complex numbers are interleaved (re,im) doubles == Julia ComplexF64 == C99 double _Complex.
 *   - dense arrays are column-major (Julia Array) with leading dimension d unless stated.
 *   - the caller owns all host buffers and must keep them alive for the duration of the call only
 *     (Julia: GC.@preserve); the library copies inputs at wae_family_create.
 *   - calls on one handle must be serialised by the caller (the reference is single-threaded).
 *   - no torch / HIP types appear in signatures; "dev" pointers are raw device addresses (uint64-castable).
 */
#ifndef WAEHIP_H
#define WAEHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wae_family wae_family;      /* opaque: all terms A_k resident in HBM + solver workspaces */

/* return codes: mapped by the host wrappers onto the reference's itsol_* flags
 * (src/NLEVP/iterative_solvers.jl:4-14), like the reference maps exceptions (:192-210). */
#define WAE_OK                 0
#define WAE_WARN_MAXITER       1   /* inner Krylov solve hit maxit on >=1 column  -> itsol_maxiter            */
#define WAE_WARN_STAGNATION    2   /*                                             -> itsol_slow_convergence   */
#define WAE_ERR_INVALID       -1   /* bad argument                                -> itsol_impossible         */
#define WAE_ERR_BREAKDOWN     -2   /* singular coarse operator / Krylov breakdown -> itsol_singular_exception */
#define WAE_ERR_EIGS          -3   /* shift-invert Arnoldi did not converge       -> itsol_arpack_exception   */
#define WAE_ERR_NAN           -4   /*                                             -> itsol_isnan              */
#define WAE_ERR_HIP           -5   /* HIP runtime failure                         -> itsol_unknown            */

/* operator applied: N = L, T = L^T, C = L^H (Julia `A'`: Householder.jl:101, iterative_solvers.jl:398,572) */
#define WAE_OP_N 0
#define WAE_OP_T 1
#define WAE_OP_C 2

/* storage orientation of the term matrices handed to wae_family_create */
#define WAE_CSC 0   /* Julia SparseMatrixCSC (colptr,rowval,nzval) */
#define WAE_CSR 1

const char *wae_last_error(void);
int wae_device_count(int *n);
/* version / build info string (static) */
const char *wae_version(void);

/* -- family ------------------------------------------------------------------------------------------
 * Upload the T term matrices of a LinearOperatorFamily once (replaces nothing in the reference by itself:
 * it is what makes `L(z)` (src/NLEVP/LinOpFam.jl:482-529) a device-resident operator instead of a new
 * SparseMatrixCSC per call).  Re-create only if a term's matrix changes, not if L.params change.
 *   d            dimension (size(L), LinOpFam.jl:385-393)
 *   T            number of terms (length(L.terms)), including the "__aux__" term if present
 *   index_bytes  4 (UInt32/Int32, Helmholtz.jl:407-408,515) or 8 (Int64)
 *   base         0 or 1 (Julia)
 *   orientation  WAE_CSC / WAE_CSR
 *   ptr[k], idx[k], val[k]  per-term arrays: ptr has d+1 entries, idx/val have nnz_k entries
 *   device       HIP device ordinal
 */
int wae_family_create(wae_family **out, int64_t d, int32_t T, int32_t index_bytes, int32_t base,
                      int32_t orientation, const void *const *ptr, const void *const *idx,
                      const double *const *val, int32_t device);
/* wae_family_create with options (opts may be NULL / nopts 0 = wae_family_create):
 *   opts[0]  symmetry tolerance of the TRANSPOSED products, default 0.  A term matrix whose transpose equals itself is stored
 *            once and applied as it is for op = T / C (`A'*y`, `A'\\b`: Householder.jl:101, iterative_solvers.jl:398,572); with 0
 *            that requires mirror entries that are equal bit for bit, so `A'` is exactly `A'`.  A finite-element matrix
 *            (`discretize`: M, K, C of src/Helmholtz.jl:405-463) is symmetric by construction but, assembled in floating point,
 *            only up to the order of its element sums; opts[0] = t > 0 accepts  |a_ij - a_ji| <= t * min(s_i, s_j),
 *            s_i = largest off-diagonal magnitude of row i (a penalty or Dirichlet diagonal entry does not widen the test),
 *            and the adjoint products of such a family then run on the same fast path as the forward ones -- differing from the
 *            exact transposed product by that assembly rounding (<= t per entry, relative to the row scale).  The Helmholtz
 *            wrappers pass 1e-14; must lie in [0, 1e-8]. */
int wae_family_create_opts(wae_family **out, int64_t d, int32_t T, int32_t index_bytes, int32_t base,
                           int32_t orientation, const void *const *ptr, const void *const *idx,
                           const double *const *val, int32_t device, const double *opts, int32_t nopts);
int wae_family_destroy(wae_family *h);
/* d, T, total nnz, and the algorithmic byte count of one spmv_sum with r right-hand sides over the terms
 * whose coefficient is non-zero in `mask` (NULL = all):  sum_k[nnz_k*(16+4)+(d+1)*4] + 2*r*d*16 */
int wae_family_info(const wae_family *h, int64_t *d, int32_t *T, int64_t *nnz_total);
int64_t wae_family_spmv_bytes(const wae_family *h, const uint8_t *mask, int32_t r);

/* -- Y = sum_k c_k op(A_k) X -------------------------------------------------------------------------
 * Replaces `L(z)*x`, `L(z,1)*x`, `L(m,n)*w`, `M*v`, `A'*y`: LinOpFam.jl:482-529 followed by a sparse
 * mat-vec (iterative_solvers.jl:307,399,571-572,581; perturbation.jl:339,352,413; Householder.jl:189-190).
 *   coeffs  T complex scalars c_k = prod_j f_kj(params; derivs) evaluated on the host (LinOpFam.jl:466-477);
 *           a term skipped by the functor (LinOpFam.jl:502-516) is passed as 0.
 *   X, Y    d x r column-major complex, host memory.  r = 0 is a no-op (`L(z)*zeros(d,0)`), and so are wae_solve /
 *           wae_solve_guess with r = 0 and wae_eig_residuals with n = 0: WAE_OK, nothing is read or written.
 */
int wae_spmv_sum(wae_family *h, const double *coeffs, const double *X, double *Y, int32_t r, int32_t op);
/* one coefficient set per column (ncoef = r): Y[:,j] = sum_k c_jk op(A_k) X[:,j] -- e.g. the residuals L(w_j) v_j of
 * all Beyn eigenpairs in one launch; ncoef = 1 is wae_spmv_sum. */
int wae_spmv_sum_cols(wae_family *h, const double *coeffs, int32_t ncoef, const double *X, double *Y, int32_t r,
                      int32_t op);
/* per-term inputs: Y = sum_k c_k A_k X_k, X = d x T column-major (regrouped perturbation recurrence,
 * SURVEY appendix C; replaces the sum over (m,n) of `L(m,n)*w` at perturbation.jl:394-415). */
int wae_spmv_sum_multi(wae_family *h, const double *coeffs, const double *X, double *Y);

/* -- solver set-up ----------------------------------------------------------------------------------
 * Build the multigrid hierarchy used to precondition every `L(z)\b` (smoothed aggregation on
 * Re(sum_k c_ref_k A_k); every term is Galerkin-projected so that coarse operators are again families).
 * Replaces the symbolic analysis UMFPACK repeats at every `\`/`lu` call (beyn.jl:65,257; perturbation.jl:329).
 * Must be called once before wae_solve / wae_beyn_moments / wae_arnoldi_shiftinvert / wae_perturb.
 * opts (may be NULL -> defaults): [0] strength threshold (0.02), [1] max coarse size (128),
 *   [2] Jacobi weight (0.8), [3] pre/post sweeps (1), [4] GMRES restart (30), [5] penalty-row ratio (1e8),
 *   [6] batch width (columns solved in lock-step, 64).
 *   [7] bit mask (as a double) of terms kept OUT of the shape matrix that the strength graph, the aggregates and the
 *       prolongator smoothing are built from (0).  Bloch families (src/Helmholtz.jl:508-513) pass the seam parts here:
 *       the solution jumps by exp(i b 2pi/N) across the seam, so no aggregate may span it.
 *   [10] weight of the POST-smoothing sweeps (0.9), [11] weight of the one sweep of the light cycle (0.5) -- the V(1,0) cycle the
 *       solves of wae_beyn_moments_rb's projected phase run, which start next to the answer and take 1-5 steps.
 *   [8], [9] workspace hints (0 = none): probe columns l and snapshot capacity of the contour integrals that will follow
 *       (wae_beyn_moments_rb): the snapshot store and the resident term products are then mapped during the set-up, behind
 *       its host work, instead of during the first integral.
 */
int wae_solver_setup(wae_family *h, const double *coeffs_ref, const double *opts, int32_t nopts);

typedef struct {
    int32_t iters_max;      /* most iterations any column needed            */
    int32_t iters_total;    /* sum over columns                              */
    int32_t n_unconverged;  /* columns that stopped at maxit                 */
    int32_t levels;         /* multigrid levels used                         */
    double  relres_max;     /* max_b ||M^-1(B_b - A X_b)|| / ||M^-1 B_b||: preconditioned (error-like) residual, recomputed */
    double  seconds;        /* wall time of the device work                  */
} wae_solve_info;

/* -- X = op(sum_k c_k A_k)^{-1} B -------------------------------------------------------------------
 * Replaces sparse `\` / `lu` + solve (UMFPACK): beyn.jl:65,257; iterative_solvers.jl:307,397-398,570-572;
 * perturbation.jl:359,423,539.  ncoef = 1: one coefficient set for all r columns;
 * ncoef = r: column j uses coeffs[j*T .. j*T+T) (independent systems solved in lock-step).
 */
int wae_solve(wae_family *h, const double *coeffs, int32_t ncoef, const double *B, double *X, int32_t r,
              int32_t op, double tol, int32_t maxit, wae_solve_info *info);

/* wae_solve with a known near-null direction per column, G[:,b].  The Newton-type solvers know the dominant direction of
 * the solution close to an eigenvalue (the current eigenvector iterate: `u = L(z)\(L(z,1)*x0)`,
 * iterative_solvers.jl:307,571-572), where the reference relies on UMFPACK factorising a nearly singular L(z).  Here the
 * direction is deflated: with u^ = M^-1 A g / ||.|| the Krylov process runs on (I - u^ u^H) M^-1 A and the solution is
 * x = x_K + alpha g, alpha cancelling the u^ component of the residual (single-level hierarchies, whose preconditioner is
 * the exact inverse, fall back to the initial guess x0 = alpha g).  G may be NULL (= wae_solve). */
int wae_solve_guess(wae_family *h, const double *coeffs, int32_t ncoef, const double *B, const double *G, double *X,
                    int32_t r, int32_t op, double tol, int32_t maxit, wae_solve_info *info);

/* -- Beyn moments -------------------------------------------------------------------------------------
 * The whole quadrature loop of `beyn` / `compute_moment_matrices` (beyn.jl:62-74,112-138,251-268):
 *   A[:,:,p] = sum_j w_j z_j^p (sum_k c_jk A_k)^{-1} V ,  p = 0..2K-1
 * npts points z[j] with effective weights w[j] (= GL weight * (b-a)/2), coefficient table npts x T
 * (row j = coefficients of L(z_j)), V d x l column-major.  Output d x l x 2K column-major (host), or, with
 * out_dev != 0, written to that device address instead (layout identical) so that the caller can reduce
 * partial moments across GPUs with RCCL before copying to the host.
 */
int wae_beyn_moments(wae_family *h, int32_t npts, const double *z, const double *w, const double *coeff_table,
                     const double *V, int32_t l, int32_t K, double tol, int32_t maxit, double *A_out,
                     uint64_t out_dev, wae_solve_info *info);

/* -- Beyn moments on several GPUs of one node, from ONE host process (SURVEY.md 8b `beyn_moments(..., ngpu)`, 8e) ------------
 * The quadrature loop of `beyn` / `compute_moment_matrices` (beyn.jl:62-74,112-138,251-268) with its points shared out over
 * ngpu devices.  handles[g]: a replica of the family on device g (wae_family_create(..., device = g) + wae_solver_setup with
 * the same arguments on each; distinct devices); the library runs one host thread and one stream per device.  nsnap > 0 (and
 * npts >= 2 nsnap): the snapshot-projection scheme of wae_beyn_moments_rb -- nsnap snapshot points solved first, split over
 * the devices by probe column when ngpu divides l (every device finishes the basis of its columns; the bases are exchanged
 * with one RCCL all-gather over xGMI), otherwise by point (raw snapshots all-gathered, every device rebuilds the basis); the
 * other points start from the projection, round-robin over the devices.  nsnap = 0: every point from a zero guess.  The
 * partial moment tensors are summed on device 0 with one RCCL reduce and copied to A_out (host, d x l x 2K column-major).
 * RCCL is loaded at first use (dlopen of librccl.so.1); ngpu = 1 runs the same code with one rank.  info: maxima / sums over
 * all devices.  Everything else as wae_beyn_moments. */
int wae_beyn_moments_mgpu(wae_family *const *handles, int32_t ngpu, int32_t npts, const double *z, const double *w,
                          const double *coeff_table, const double *V, int32_t l, int32_t K, double tol, int32_t maxit,
                          int32_t nsnap, double *A_out, wae_solve_info *info);

/* -- residual check of eigenpairs (the last step of `beyn`'s callers: which Ritz pairs are eigenpairs) --------------
 * res_out[j] = || sum_k c_jk A_k v_j || / sum_k |c_jk| || A_k v_j ||   for the n pairs (coeff_table: n x T complex, row j =
 * the coefficients of L(omega_j); v_j = column j of the column-major d x n matrix P on the host, or P_dev on the device).
 * This componentwise-scaled backward error is meaningful in the presence of penalty rows (1e15-sized admittance entries,
 * src/Helmholtz.jl:151-156), where ||L(omega) v|| / ||v|| is not. */
int wae_eig_residuals(wae_family *h, int32_t n, const double *coeff_table, const double *P, uint64_t P_dev, double *res_out);

/* -- Beyn moments with snapshot-projection initial guesses -----------------------------------------------------
 * The solutions X(z) = L(z)^{-1} V along a contour form a low-dimensional manifold (a handful of poles near the
 * contour plus a smooth part) -- the observation behind the reference's `generate_subspace`/`project`
 * (src/NLEVP/beyn.jl:429-560).  Here it accelerates the integrand evaluation of `beyn`/`compute_moment_matrices`
 * (src/NLEVP/beyn.jl:62-71,253-259) itself, without changing its result: a few quadrature points are solved from a
 * zero guess and kept as snapshots (mode 0); for all other points (mode 1) every system starts from the Galerkin
 * projection of its solution on the per-column span of the snapshots and multigrid-GMRES only has to supply the rest,
 * to the same tolerance relative to the same right-hand side.
 *   nbasis: capacity of the snapshot store in snapshots (each d x l complex, interleaved [row][column]).
 *   mode 0: solve the npts points, accumulate their moment contributions and append their solutions to the store
 *           (slot0 = number of snapshots already there; 0 starts a new basis).  Progressive: a chunk of points starts
 *           from the projection on the snapshots taken before it, so pass the points in a spread-out order.
 *   mode 1: the store holds slot0 raw snapshots (e.g. all-gathered from several GPUs): orthonormalise them per
 *           column (in place), project every term, then process the npts points with projected initial guesses.
 *   mode 2: as mode 1 with the basis as mode 0 calls left it (no rebuild).
 *   mode 3: solve the npts points from ZERO guesses, accumulate their moment contributions and write their solutions RAW into the
 *           store slots slot0 .. slot0+npts-1; the handle's basis is not touched.  (A multi-GPU rank's share of the snapshot
 *           points as full-width batches; another rank -- or mode 4 -- builds the basis.)
 *   mode 4: the store holds slot0 raw snapshots of THIS call's l columns: orthonormalise them per column (in place) and project
 *           every term that has a non-zero coefficient in the table (npts rows) -- then return: no system is solved, the
 *           moments are not touched.  With mode 3 and the exchanges of wae_rb_export / wae_rb_import this is the "hybrid" split
 *           of the snapshot phase over G GPUs: points for the solves, probe columns for the basis.
 *   (Environment WAE_RB_ENRICH=<n>: in modes 1/2 append a chunk of points that still needed more than n iterations to
 *   the basis while the store has room.  Off by default: it did not pay on the benchmark contour.)
 *   Q_dev : device pointer of the snapshot store, or 0 for a store owned by the handle; a caller-owned store is
 *           what a multi-GPU driver all-gathers between modes 0 and 1.
 *   accumulate != 0: add to the moments already in out_dev instead of zeroing them first (requires out_dev).
 *   V     : in mode 2 V may be NULL: the probe matrix of the mode 0/1 call that started the basis (kept on the device) is
 *           used again, which saves the second host-to-device copy of a pass.  Not after wae_rb_import.
 * Everything else as wae_beyn_moments.
 *   l_total, col0: the moment tensor has l_total columns and V holds columns col0 .. col0+l-1 of the probe matrix
 *           (l_total <= 0: l_total = l, col0 = 0).  A multi-GPU driver lets every rank take ALL snapshot points for its
 *           own slice of the probe columns (mode 0 stays progressive and leaves a finished basis for that slice), then
 *           exchanges the bases: wae_rb_export / all-gather / wae_rb_import, and runs mode 2 on its share of the points. */
int wae_beyn_moments_rb(wae_family *h, int32_t npts, const double *z, const double *w, const double *coeff_table, const double *V,
                        int32_t l, int32_t K, double tol, int32_t maxit, int32_t mode, int32_t nbasis, int32_t slot0, uint64_t Q_dev,
                        double *A_out, uint64_t out_dev, int32_t accumulate, int32_t l_total, int32_t col0, wae_solve_info *info);
/* The snapshot basis of the handle, host side: S vectors per column, l columns, nk projected terms kact[0..nk-1];
 * Hk dense [ki][s][i][c] (= q_i^H A_k q_s of column c's basis, c fastest), g [i][c] (= q_i^H v_c); complex interleaved.
 * export: pass NULL arrays to query the sizes first.  import: installs a basis whose vectors lie in Q_dev
 * (S x d x l, interleaved [row][column], orthonormal per column) for mode 2. */
int wae_rb_export(wae_family *h, int32_t *S_out, int32_t *l_out, int32_t *nk_out, int32_t *kact_out, double *Hk_out, double *g_out);
int wae_rb_import(wae_family *h, int32_t S, int32_t l, uint64_t Q_dev, int32_t nk, const int32_t *kact, const double *Hk, const double *g);

/* -- shift-invert Arnoldi factorisation for (A, M), A = sum cA_k A_k, M = sum cM_k A_k -----------------
 * The device half of `Arpack.eigs(A,M,nev=nev,sigma=0,v0=v0)` and of the adjoint call on (A',M')
 * (Householder.jl:100-101, iterative_solvers.jl:132-133):  m steps of Arnoldi on  op(A)^{-1} op(M)
 * started from v0, every step one multigrid-GMRES solve on the device:
 *      op(A)^{-1} op(M) V[:,0:m] = V[:,0:m+1] H ,   V^H V = I.
 * The small (m x m) Hessenberg eigenproblem, Ritz extraction and restarts stay on the host side
 * (Julia LinearAlgebra / numpy), as ARPACK's do.  op = WAE_OP_N (right) or WAE_OP_C (left: A^H, M^H).
 *   H_out  (m+1) x m column-major complex;  V_out  d x (m+1) column-major complex.
 *   If an invariant subspace ends the recurrence after j < m steps, the remaining columns of H_out / V_out
 *   are zero (H[j+1,j] = 0 marks the end) and the call still returns WAE_OK.
 */
int wae_arnoldi_shiftinvert(wae_family *h, const double *coeffsA, const double *coeffsM, int32_t m,
                            const double *v0, int32_t op, double tol, int32_t maxit, double *H_out,
                            double *V_out, wae_solve_info *info);
/* The same for nsys operator pairs at once, all Arnoldi processes advancing in lock-step (one batched solve per step):
 * refining all the estimates a Beyn solve returned costs about as much as refining one, because a single-column solve
 * is latency-bound.  coeffsA, coeffsM: nsys x T; v0: d x nsys column-major; H_out: nsys blocks of (m+1) x m;
 * V_out: nsys blocks of d x (m+1), column-major.  A column whose Krylov space becomes invariant stops (its later H
 * entries and basis vectors are zero).  ritz_tol > 0: stop as soon as the dominant Ritz pair of every process has a
 * relative residual |h_{k+1,k}| |y_k| / |theta| <= ritz_tol (the H columns of the steps not taken are zero, their basis
 * vectors in V_out are NOT written -- 128 MB of host memory each at 1M DoF and 8 systems; pass zeroed or scratch storage);
 * the inner solves of the later steps are then relaxed as the Ritz residual falls (inexact Arnoldi: step k is solved to
 * tol / (10 x relative Ritz residual after step k-1), at most 1e-3).  0: always m steps, every solve to tol. */
int wae_arnoldi_shiftinvert_batch(wae_family *h, int32_t nsys, const double *coeffsA, const double *coeffsM, int32_t m, const double *v0,
                                  int32_t op, double tol, int32_t maxit, double ritz_tol, double *H_out, double *V_out,
                                  wae_solve_info *info);

/* -- adjoint perturbation recurrence -----------------------------------------------------------------
 * Replaces `perturb` / `perturb_disk` / `perturb_norm` (perturbation.jl:319-367,374-444,487-560) for a
 * two-parameter expansion L(m,n) = d^m/dλ^m d^n/dε^n L /(m! n!):
 *   coeff_table[(m*(N+1)+n)*T + k]  = coefficient of term k in L(m,n), m,n = 0..N  (0 where m+n>N)
 *   v0, v0adj   base eigenvectors (un-normalised as the reference receives them)
 *   norm_mode   0: perturb (no `c` normalisation)  1: perturb_disk  2: perturb_norm with Y = sum cY_k A_k;
 *               +16: eigenvalue series only (skip the solve at order N; what householder/mslp need, Householder.jl:115-116)
 * Outputs lambda_out[N+1] (entry 0 untouched, the wrappers overwrite it: LinOpFam.jl:555), v_out d x (N+1).
 */
int wae_perturb(wae_family *h, const double *coeff_table, int32_t N, const double *v0, const double *v0adj,
                int32_t norm_mode, const double *coeffsY, double tol, int32_t maxit, double *lambda_out,
                double *v_out, wae_solve_info *info);

/* -- device-resident multivectors ("slots") for the Newton-type solvers ------------------------------------------
 * `householder` (Householder.jl:70-192) iterates, per start value, on a right and a left eigenvector estimate: every Newton step
 * runs two shift-invert Arnoldi processes from them (Householder.jl:100-101), forms the Ritz vectors, and feeds both to the
 * perturbation step (Householder.jl:115-116, perturbation.jl:319-367).  Through wae_arnoldi_shiftinvert_batch / wae_perturb those
 * vectors cross the host boundary five times per step; with slots they stay in HBM from the first step to the last.
 * A family owns WAE_NSLOTS slots; a slot holds d x ncols complex numbers (column-major for the caller, stored in the library's row
 * numbering).  All arrays of column indices are 0-based.
 *   wae_slot_write   (re)creates the slot with ncols_total columns if it has a different column count (new columns are zero) and
 *                    copies X (d x ncols, column-major, host) into columns col0 .. col0+ncols-1.  ncols = 0: create / resize only.
 *   wae_slot_read    copies columns col0 .. col0+ncols-1 to X (host).
 *   wae_slot_axpby   dst[:, dst_cols[i]] = alpha[i] * src[:, src_cols[i]] + beta[i] * dst[:, dst_cols[i]],  i < n, one after the other
 *                    (alpha, beta: n complex numbers; src and dst may be the same slot and column: a scaling; conj_src != 0: the
 *                    conjugate of the source column is used).  The relaxed update of Householder.jl:173-176; with conj_src the left
 *                    start vectors conj(v0) of Householder.jl:84-86 from the right ones.
 *   wae_slot_forms   out[i] = a_i^H op(sum_k coeffs[i][k] A_k) b_i  for n pairs of columns a_i = slot a[:, a_cols[i]], b_i likewise
 *                    (coeffs: n x T): the normalisations v^H M v and v_adj^H L'(z) v of Householder.jl:189-190 without moving a vector.
 *   wae_arnoldi_shiftinvert_slots   wae_arnoldi_shiftinvert_batch with the start vectors taken from slot columns v0_cols[0..nsys-1] and
 *                    the basis KEPT on the device (only H_out comes back): the caller solves the small Hessenberg eigenproblems and
 *   wae_arnoldi_ritz_to_slot   writes  sum_j y[s][j] v_j^(s)  (y: nsys x ny complex, ny <= steps taken + 1; normalise != 0: scaled to
 *                    unit 2-norm) of the basis of the LAST wae_arnoldi_shiftinvert_slots call (same nsys) into dst[:, dst_cols[s]].
 *   wae_perturb_slots   wae_perturb with v0 = slot v[:, v_col], v0adj = slot vadj[:, vadj_col]; v_out may be NULL (eigenvalue series
 *                    only: what householder / mslp need).
 * Both Arnoldi entries: with ritz_tol > 0 a start column that is not close to an eigenvector (||M^-1 op(A) v0|| > 0.1 ||v0||, M^-1 the
 * multigrid cycle: known before the first solve) is first replaced by one step of inverse iteration from it, solved to 1e-3 -- the
 * process then starts from that vector (V[:,0] is the replaced start). */
#define WAE_NSLOTS 8
int wae_slot_write(wae_family *h, int32_t slot, int32_t ncols_total, int32_t col0, int32_t ncols, const double *X);
int wae_slot_read(wae_family *h, int32_t slot, int32_t col0, int32_t ncols, double *X);
int wae_slot_axpby(wae_family *h, int32_t n, int32_t dst_slot, const int32_t *dst_cols, int32_t src_slot, const int32_t *src_cols,
                   const double *alpha, const double *beta, int32_t conj_src);
int wae_slot_forms(wae_family *h, int32_t n, const double *coeffs, int32_t op, int32_t a_slot, const int32_t *a_cols, int32_t b_slot,
                   const int32_t *b_cols, double *out);
int wae_arnoldi_shiftinvert_slots(wae_family *h, int32_t nsys, const double *coeffsA, const double *coeffsM, int32_t m, int32_t v0_slot,
                                  const int32_t *v0_cols, int32_t op, double tol, int32_t maxit, double ritz_tol, double *H_out,
                                  wae_solve_info *info);
int wae_arnoldi_ritz_to_slot(wae_family *h, int32_t nsys, int32_t ny, const double *y, int32_t dst_slot, const int32_t *dst_cols,
                             int32_t normalise);
int wae_perturb_slots(wae_family *h, const double *coeff_table, int32_t N, int32_t v_slot, int32_t v_col, int32_t vadj_slot, int32_t vadj_col,
                      int32_t norm_mode, const double *coeffsY, double tol, int32_t maxit, double *lambda_out, double *v_out,
                      wae_solve_info *info);

/* -- P1 assembly on the device (input production) ------------------------------------------------
 * Mass and stiffness matrices of the P1 tetrahedral discretisation, as `discretize` assembles them for the "interior"
 * domain (src/Helmholtz.jl:405-441 with the element kernels src/FEM/FEM.jl:704-710,1745-1766):
 *     M_ab += |det J|/120 (1 + delta_ab),      K_ab += -c_tet^2 |det J|/6 grad(phi_a).grad(phi_b)
 * points: 3 doubles per point (x,y,z); tets: 4 point indices (0-based) per tetrahedron; c_tet: speed of sound per
 * tetrahedron (NULL = 1).  Triplets are sorted and summed on the device (hipCUB), deterministic, no atomics.  The two
 * matrices share one CSR pattern (rowptr npoints+1, col nnz; real values).  wae_p1_assemble returns a handle, wae_p1_info
 * the sizes, wae_p1_get copies the arrays out (any pointer may be NULL), wae_p1_free releases it. */
int wae_p1_assemble(int32_t device, int64_t npoints, const double *points, int64_t ntets, const int32_t *tets, const double *c_tet, void **out);
/* The other two operators of `discretize` for a P1 Helmholtz problem, same pipeline and same handle type (wae_p1_info /
 * wae_p1_get / wae_p1_free; the values come back in the `mass` array of wae_p1_get, `stiff` is zero):
 *  - admittance boundary (src/Helmholtz.jl:443-463 with src/FEM/FEM.jl:9-20,435-441): per boundary triangle
 *        b_ab = c_tri |(x0-x2) x (x1-x2)| (1 + delta_ab) / 24 ;   the operator term is  C = -i b  (Helmholtz.jl:459).
 *    tris: 3 point indices (0-based) per triangle, c_tri: speed of sound of the tetrahedron behind each (NULL = 1).
 *  - flame (src/Helmholtz.jl:292-344,464-487 with FEM.jl:2429-2431,2442-2448): Q = sum over the flame tetrahedra of S (x) g,
 *        S_a = |det J|/24 on the four nodes of a flame tetrahedron,   g_b = -nlocal grad(phi_b).n_ref on the reference
 *        tetrahedron,   nlocal = nglobal_scaled / V_flame  (the caller passes (gamma-1)/rho * Q02U0, Helmholtz.jl:325; the
 *        flame volume is summed on the device and returned in volume_out if not NULL).
 *    flame_tets: indices into tets of the nflame flame tetrahedra; ref_tet: index of the tetrahedron that contains the
 *    reference point (Meshutils.jl:800-816 finds it; that search stays on the host); n_ref: 3 doubles. */
int wae_p1_assemble_boundary(int32_t device, int64_t npoints, const double *points, int64_t ntris, const int32_t *tris, const double *c_tri, void **out);
int wae_p1_assemble_flame(int32_t device, int64_t npoints, const double *points, int64_t ntets, const int32_t *tets, int64_t nflame,
                          const int32_t *flame_tets, int32_t ref_tet, const double *n_ref, double nglobal_scaled, void **out, double *volume_out);
int wae_p1_info(const void *handle, int64_t *npoints, int64_t *nnz);
int wae_p1_get(const void *handle, int32_t *rowptr, int32_t *col, double *mass, double *stiff);
int wae_p1_free(void *handle);
/* Discrete-adjoint shape sensitivity (src/shape_sensitivity.jl:16-141) of an eigenvalue w.r.t. the coordinates of surface
 * points, for the interior (M, K) and the admittance-boundary (w*Y*C) parts of the P1 Helmholtz operator.  As in the
 * reference the operator derivative is a central difference (step h) of two local re-discretisations of the simplices
 * that touch the point; here one device thread does that for one (point, simplex, coordinate).
 *   pair_pt_t[i], pair_tet[i]: surface point and one tetrahedron containing it (npair_t pairs); pair_pt_s, pair_tri:
 *   the same for boundary triangles (tris: 3 point indices each, c_tri: speed of sound of the adjacent tetrahedron).
 *   omega: eigenvalue (re, im); omegaY: omega*Y (re, im); v, v_adj: eigenvectors, normalised v'v = 1, v_adj' L'(omega) v = 1.
 *   out_t[3*npair_t], out_s[3*npair_s] (complex): -v_adj' (dL/dx) v contribution of every pair and coordinate; the
 *   caller sums them per point (deterministic order). */
int wae_p1_shape_sensitivity(int32_t device, int64_t npoints, const double *points, const int32_t *tets, const double *c_tet, int64_t npair_t,
                             const int32_t *pair_pt_t, const int32_t *pair_tet, const int32_t *tris, const double *c_tri, int64_t npair_s,
                             const int32_t *pair_pt_s, const int32_t *pair_tri, int64_t ntets, int64_t ntris, const double *omega,
                             const double *omegaY, const double *v, const double *v_adj, double h, double *out_t, double *out_s);

/* Flame part of the same sensitivity (a :flame entry in dscrp; shape_sensitivity.jl:62-141 with Helmholtz.jl:292-344,464-487):
 * per (surface point, flame tetrahedron touching it) pair and coordinate, |det J| of the tetrahedron with the point moved by +h
 * and -h (det_pm[pair][3][2]) and, per pair, the sum of conj(v_adj) over the tetrahedron's nodes (ssum, complex); per listed
 * vertex of the reference tetrahedron (pair_pt_r) and coordinate, sum_b (grad(phi_b).n_ref) v_b on the reference tetrahedron with
 * that vertex moved by +h / -h (g_pm[pair][3][2], complex) and the undisplaced value g0.  The caller (helmholtz/assemble.py,
 * julia) sums the pairs of a point in order and forms  -v_adj' (Q+ - Q-)/(2h) v  with Q = S (x) g,  S_a = |det J|/24,
 * g_b = -(nglobal_scaled / volume of the point's flame tetrahedra) grad(phi_b).n_ref  -- the reference re-discretises the flame
 * domain REDUCED to the simplices at the point, volume included. */
int wae_p1_shape_sensitivity_flame(int32_t device, int64_t npoints, const double *points, int64_t ntets, const int32_t *tets, int64_t npair,
                                   const int32_t *pair_pt, const int32_t *pair_tet, int32_t ref_tet, int64_t npair_r, const int32_t *pair_pt_r,
                                   const double *n_ref, const double *v, const double *v_adj, double h, double *det_pm, double *ssum,
                                   double *g_pm, double *g0);

/* -- measurement helpers (bench.py) --------------------------------------------------------------------
 * Time `reps` launches of the fused multi-term SpMV on device-resident data with HIP events on the
 * library's own stream; r right-hand sides.  ms_out = average milliseconds per launch. */
int wae_bench_spmv(wae_family *h, const double *coeffs, int32_t r, int32_t reps, double *ms_out);
/* the same for an operator of the multigrid hierarchy (which = 0: level operator `level`, 1: restriction from `level`, 2: prolongation
 * to `level`, an in-place update of the fine vector), with its
 * algorithmic bytes per launch (SURVEY 8d layout: 16 + 4 bytes per nonzero and plane with a non-zero coefficient -- 8 + 4 for the
 * real restriction -- row pointers, input and output vectors touched once): the roofline line of the level-1 kernels. */
int wae_bench_spmv_level(wae_family *h, const double *coeffs, int32_t which, int32_t level, int32_t r, int32_t reps, double *ms_out,
                         int64_t *bytes_out);
/* device triad a = b + s*c over n doubles: measured streaming bandwidth in GB/s (the best of five grid sizes: the rate depends on the
 * shape of the launch by up to 20 %) */
int wae_bench_triad(int32_t device, int64_t n, int32_t reps, double *gbs_out);

/* -- test hook (tests/ only) ----------------------------------------------------------------------------
 * The solver applies the operator `L(z)*X` (LinOpFam.jl:482-529) in fused forms that no public entry exposes -- residual,
 * damped-Jacobi sweep, product + first sweep, converged-chunk masks -- and on the coarse levels of its hierarchy.  This entry
 * runs ONE such launch so that the parity tests can compare every form with the CPU oracle:
 *   which = 0: the operator of multigrid level `level` (0 = the family itself; >= 1 needs wae_solver_setup);
 *   which = 1: the restriction from `level` to `level + 1` (coefficients ignored; mode 0);
 *   which = 2: the prolongation from `level + 1` to `level`, Y = B + P X (coefficients ignored; mode 3).
 *   mode: 0 Y = A X | 1 Y = B - A X | 2 Y = X + w/diag (B - A X) | 3 Y = B + A X | 4 Y = (A X)/diag | 5 Y = (B - A X)/diag
 *         | 6 Y = A X and B2 = w/diag (A X)  (diag = the diagonal of sum_k c_k A_k, w = jac_w)
 *   coeffs: ncoef x T (ncoef = 1: one system; ncoef = r: one coefficient row per column);
 *   X, B, Y, B2: column-major n_in x r / n_out x r complex (B may be NULL for modes 0, 4; B2 only for mode 6).  Level 0 is in
 *   the caller's row numbering; coarser levels in the hierarchy's own (their size: n_in/n_out = 0 on entry returns it in *n_out_q).
 *   cmask: NULL or one byte per 8-column chunk; 0 = the chunk is skipped and keeps the values Y (and B2) hold on entry.
 *   flags bit 0: bypass the tile-local storage (the plain CSR kernels), for A/B comparisons of the two storage forms. */
int wae_debug_spmv(wae_family *h, int32_t which, int32_t level, int32_t mode, const double *coeffs, int32_t ncoef, const double *X,
                   const double *B, double *Y, double *B2, int32_t r, int32_t op, double jac_w, const uint8_t *cmask, int32_t flags,
                   int64_t *n_in_q, int64_t *n_out_q);

#ifdef __cplusplus
}
#endif
#endif
