/* abi_caller.c -- a plain C99 caller of libwaehip.so, compiled with gcc against include/waehip.h by the tests.
 *
 * It does what a foreign host (Julia's ccall) does with the library: hands over the term matrices as
 * 1-based CSC arrays with UInt32 or Int64 indices (Julia SparseMatrixCSC; src/Helmholtz.jl:407-408,515), then
 *     wae_family_create -> wae_solver_setup -> wae_beyn_moments -> wae_family_destroy
 * and writes the moment tensor and the solve statistics to a file the test compares with the CPU oracle.
 *
 *   abi_caller layout                 print sizeof/offsetof of wae_solve_info (no library call; CPU test)
 *   abi_caller run <in.bin> <out.bin> run the sequence above (GPU test)
 *
 * in.bin  (little endian): int64 d, T, index_bytes, base, npts, l, K, maxit; double tol;
 *          per term: int64 nnz; ptr[d+1], idx[nnz] (index_bytes each); val[2*nnz] doubles;
 *          coeffs_ref[2T]; z[2 npts]; w[2 npts]; coeff_table[2 npts T]; V[2 d l]   (doubles)
 * out.bin: int64 rc_create, rc_setup, rc_moments, rc_destroy; wae_solve_info as 4 int64 + 2 doubles; A[2 d l 2K] doubles
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "waehip.h"

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "out of memory\n"); exit(3); }
    return p;
}
static void xread(void *dst, size_t size, size_t count, FILE *f) {
    if (fread(dst, size, count, f) != count) { fprintf(stderr, "short read\n"); exit(3); }
}

int main(int argc, char **argv) {
    if (argc >= 2 && strcmp(argv[1], "layout") == 0) {
        printf("sizeof %zu iters_max %zu iters_total %zu n_unconverged %zu levels %zu relres_max %zu seconds %zu\n",
               sizeof(wae_solve_info), offsetof(wae_solve_info, iters_max), offsetof(wae_solve_info, iters_total),
               offsetof(wae_solve_info, n_unconverged), offsetof(wae_solve_info, levels), offsetof(wae_solve_info, relres_max),
               offsetof(wae_solve_info, seconds));
        printf("codes %d %d %d %d %d %d %d %d ops %d %d %d orient %d %d\n", WAE_OK, WAE_WARN_MAXITER, WAE_WARN_STAGNATION, WAE_ERR_INVALID,
               WAE_ERR_BREAKDOWN, WAE_ERR_EIGS, WAE_ERR_NAN, WAE_ERR_HIP, WAE_OP_N, WAE_OP_T, WAE_OP_C, WAE_CSC, WAE_CSR);
        return 0;
    }
    if (argc != 4 || strcmp(argv[1], "run") != 0) {
        fprintf(stderr, "usage: %s layout | run <in.bin> <out.bin>\n", argv[0]);
        return 2;
    }
    FILE *f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 2; }
    int64_t hd[8];
    double tol;
    xread(hd, sizeof(int64_t), 8, f);
    xread(&tol, sizeof(double), 1, f);
    const int64_t d = hd[0], T = hd[1], ib = hd[2], base = hd[3], npts = hd[4], l = hd[5], K = hd[6], maxit = hd[7];
    void **ptr = (void **)xmalloc((size_t)T * sizeof(void *));
    void **idx = (void **)xmalloc((size_t)T * sizeof(void *));
    double **val = (double **)xmalloc((size_t)T * sizeof(double *));
    for (int64_t k = 0; k < T; ++k) {
        int64_t nnz;
        xread(&nnz, sizeof(int64_t), 1, f);
        ptr[k] = xmalloc((size_t)(d + 1) * (size_t)ib);
        idx[k] = xmalloc((size_t)nnz * (size_t)ib);
        val[k] = (double *)xmalloc((size_t)nnz * 2 * sizeof(double));
        xread(ptr[k], (size_t)ib, (size_t)(d + 1), f);
        xread(idx[k], (size_t)ib, (size_t)nnz, f);
        xread(val[k], sizeof(double), (size_t)nnz * 2, f);
    }
    double *cref = (double *)xmalloc((size_t)T * 2 * sizeof(double));
    double *z = (double *)xmalloc((size_t)npts * 2 * sizeof(double));
    double *w = (double *)xmalloc((size_t)npts * 2 * sizeof(double));
    double *ct = (double *)xmalloc((size_t)npts * (size_t)T * 2 * sizeof(double));
    double *V = (double *)xmalloc((size_t)d * (size_t)l * 2 * sizeof(double));
    xread(cref, sizeof(double), (size_t)T * 2, f);
    xread(z, sizeof(double), (size_t)npts * 2, f);
    xread(w, sizeof(double), (size_t)npts * 2, f);
    xread(ct, sizeof(double), (size_t)npts * (size_t)T * 2, f);
    xread(V, sizeof(double), (size_t)d * (size_t)l * 2, f);
    fclose(f);

    const size_t acnt = (size_t)d * (size_t)l * 2 * (size_t)K * 2;
    double *A = (double *)xmalloc(acnt * sizeof(double));
    memset(A, 0, acnt * sizeof(double));
    int64_t rc[4] = {99, 99, 99, 99};
    wae_solve_info info;
    memset(&info, 0, sizeof(info));
    wae_family *h = NULL;
    int ndev = 0;
    if (wae_device_count(&ndev) != WAE_OK || ndev < 1) { fprintf(stderr, "no HIP device: %s\n", wae_last_error()); return 4; }
    rc[0] = wae_family_create(&h, d, (int32_t)T, (int32_t)ib, (int32_t)base, WAE_CSC, (const void *const *)ptr, (const void *const *)idx,
                              (const double *const *)val, 0);
    if (rc[0] != WAE_OK) fprintf(stderr, "create: %s\n", wae_last_error());
    if (rc[0] == WAE_OK) {
        int64_t dd = 0, nnzt = 0;
        int32_t TT = 0;
        if (wae_family_info(h, &dd, &TT, &nnzt) != WAE_OK || dd != d || TT != T) { fprintf(stderr, "family_info mismatch\n"); return 5; }
        rc[1] = wae_solver_setup(h, cref, NULL, 0);
        if (rc[1] != WAE_OK) fprintf(stderr, "setup: %s\n", wae_last_error());
    }
    if (rc[1] == WAE_OK) {
        rc[2] = wae_beyn_moments(h, (int32_t)npts, z, w, ct, V, (int32_t)l, (int32_t)K, tol, (int32_t)maxit, A, 0, &info);
        if (rc[2] < 0) fprintf(stderr, "moments: %s\n", wae_last_error());
    }
    rc[3] = wae_family_destroy(h);
    /* a call on a bad argument must come back as a code with a message, never as a crash */
    if (wae_family_create(NULL, d, (int32_t)T, 4, 1, WAE_CSC, NULL, NULL, NULL, 0) != WAE_ERR_INVALID || strlen(wae_last_error()) == 0) {
        fprintf(stderr, "bad-argument call did not return WAE_ERR_INVALID\n");
        return 6;
    }
    f = fopen(argv[3], "wb");
    if (!f) { perror(argv[3]); return 2; }
    int64_t ii[4] = {info.iters_max, info.iters_total, info.n_unconverged, info.levels};
    double dd2[2] = {info.relres_max, info.seconds};
    fwrite(rc, sizeof(int64_t), 4, f);
    fwrite(ii, sizeof(int64_t), 4, f);
    fwrite(dd2, sizeof(double), 2, f);
    fwrite(A, sizeof(double), acnt, f);
    fclose(f);
    printf("abi_caller: rc %lld %lld %lld %lld  iters_total %d  unconverged %d  levels %d  relres %.2e  (%s)\n", (long long)rc[0], (long long)rc[1],
           (long long)rc[2], (long long)rc[3], info.iters_total, info.n_unconverged, info.levels, info.relres_max, wae_version());
    return 0;
}
