// Internal declarations shared by the HIP translation units of libgpbo (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpbo.h"

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

#define GPBO_NB 64 /* Cholesky / triangular-inverse block size */
#define GPBO_CHOLINV_MAX_NP 32768 /* largest padded size the fused sweep's plan covers (cholinv_plan.h: 32-bit tile offsets) */
#ifndef GPBO_KS_SLICE
#define GPBO_KS_SLICE 64 /* observations per workgroup of the K(X*,X) kernel = rows per mu_part slice */
#endif

#define GPBO_CHECK_LAUNCH()                                  \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return GPBO_ERR_LAUNCH; \
    } while (0)

static inline hipStream_t gpbo_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// fp64 MFMA 16x16x4: lane l supplies A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15];
// result register r of lane l is C[row = (l>>4) + 4r][col = l&15].
__device__ __forceinline__ d4_t mfma_f64_16x16x4(double a, double b, d4_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// __syncthreads() with the wave's own outstanding LDS and memory operations waited for EXPLICITLY.  hipcc (ROCm 7.2, gfx950)
// was seen to emit the s_barrier of a __syncthreads() at a LOOP HEADER without the s_waitcnt lgkmcnt(0) in front of it
// (nlml_grid_kernel's elimination loop: the last ds_write of step c still in flight when another wave, released by the
// barrier, read that entry in step c + 1): 2 % of the cells of a 2,600-cell launch at d = 16 came out wrong, differently
// every run, found by tools/fuzz_ard.py in round 5.  The wait is two scalar instructions; use this wherever a barrier
// publishes LDS (or memory) writes of the loop body before it.
__device__ __forceinline__ void gpbo_syncthreads() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

// (value, index) order of every arg-max reduction: larger value first, ties to the LOWER index (point_selector.py:207)
__device__ __forceinline__ bool gpbo_better(double v2, int64_t i2, double v, int64_t i) {
    return (v2 > v) || (v2 == v && i2 < i);
}

// acquisition value from the posterior mean and standard deviation (point_selector.py:204; EI: SURVEY.md 8 a10)
__device__ __forceinline__ double gpbo_acquisition(int kind, double mu, double sigma, double p0, double p1) {
    if (kind == GPBO_ACQ_LCB) return p0 * sigma - mu;
    // Expected improvement for minimisation: imp = f_best - mu - xi
    const double imp = p0 - mu - p1;
    if (!(sigma > 0.0)) return (sigma == 0.0) ? fmax(imp, 0.0) : sigma;  // sigma NaN propagates
    const double z = imp / sigma;
    const double cdf = 0.5 * erfc(-z * 0.70710678118654752440);
    const double pdf = exp(-0.5 * z * z) * 0.39894228040143267794;
    return imp * cdf + sigma * pdf;
}

// The same acquisition as an UPPER BOUND (the exact prefix bound: mu reported from below, sigma from above): a value that is
// >= what gpbo_acquisition COMPUTES for any (mu' >= mu, sigma' <= sigma), roundings included.
//   LCB (explore >= 0 on this route): p0 * sigma - mu is monotone operation by operation - the plain form is the bound.
//   EI: imp Phi(z) + sigma phi(z) cancels for z << 0 (the two terms agree to 1 / z^2) and erfc / exp amplify the rounding of
//   their arguments by ~ 2 z^2: the computed value carries an absolute error of up to ~ (c + 4 z^2) eps (|t1| + t2).  That
//   bound is evaluated HERE, at the bound's own point - g(z) = (64 + 4 z^2) phi(z) decreases in |z| and sigma phi(z) grows
//   with sigma and with imp, so it dominates the error of the plain pass's evaluation at (mu', sigma') as well - and added
//   twice (once for each of the two evaluations being compared), plus a floor for results that underflow.
__device__ __forceinline__ double gpbo_acquisition_ub(int kind, double mu, double sigma, double p0, double p1) {
    if (kind == GPBO_ACQ_LCB) return p0 * sigma - mu;
    const double imp = p0 - mu - p1;
    if (!(sigma > 0.0)) return (sigma == 0.0) ? fmax(imp, 0.0) : sigma;  // sigma NaN propagates
    const double z = imp / sigma;
    const double cdf = 0.5 * erfc(-z * 0.70710678118654752440);
    const double pdf = exp(-0.5 * z * z) * 0.39894228040143267794;
    const double t1 = imp * cdf, t2 = sigma * pdf;
    const double pad = 2.0 * (64.0 + 4.0 * z * z) * 1.1102230246251565e-16 * (fabs(t1) + t2);
    return (t1 + t2) + pad + 1e-300;
}

// launchers implemented in the individual .hip files (host side, enqueue only)
// prefix bound: variance floor / pad (see sigma_acq_kernel's epilogue)
#define GPBO_BOUND_VAR_PAD 1e-8
// K(X*,X) with the distances on the matrix cores (kstar_mfma.hip): prefix-bound route only
int64_t gpbo_kstar_mfma_prep_bytes(int64_t Np);
int gpbo_kstar_mfma_slice(int64_t Np);
int gpbo_kstar_mfma_prep(const double *X, int64_t N, int64_t Np, int32_t d, const double *ls_host, const double *alpha,
                         void *prep_buf, void *stream);
int gpbo_kstar_mu_mfma(const double *Xs, int64_t Mc, int64_t N, int64_t Np, int32_t d, const double *ls_host,
                       const double *alpha, const void *prep_buf, double *KsT, int64_t ldk, double *mu_part,
                       int64_t store_rows, void *stream);
int gpbo_kstar_mu_rows(const double *Xs, int64_t Mc, const double *Xsc, int64_t N, int64_t Np, int32_t d,
                       const double *ls_host, const double *alpha, double diag_add, int64_t cand_base, double *KsT,
                       int64_t ldk, double *mu_part, int64_t store_rows, void *stream);
int gpbo_kstar_mu_anyd(const double *Xs, int64_t Mc, const double *X, int64_t N, int64_t Np, int32_t d, const double *ls_host,
                       const double *alpha, double diag_add, int64_t cand_base, double *KsT, int64_t ldk, double *mu_part,
                       void *stream);
int gpbo_kstar_mu_mixed(const double *Xs, int64_t Mc, const double *Xsc, int64_t N, int64_t Np, int32_t d,
                        const double *ls_host, const double *alpha, double diag_add, int64_t cand_base, float *KsT,
                        int64_t ldk, double *mu_part, void *stream);
int gpbo_kxx_launch(const double *X, int64_t N, int32_t d, const double *ls_host, double jitter1, double jitter2,
                    double *Kp, int64_t Np, double *K2, int64_t ld2, int32_t *info0, void *stream);
// fused Cholesky + inverse factor on the stacked matrix [A | W] (cholinv.hip)
int gpbo_cholinv_run(double *S, int64_t ld, int64_t Np, int32_t *info, const int *opt, hipStream_t st);
int gpbo_launch_transpose_w(const double *W, int64_t ldw, int64_t Np, double *U, hipStream_t st);
int gpbo_scale_points_launch(const double *X, int64_t N, int64_t Np, int32_t d, const double *ls_host, double *Xsc,
                             unsigned long long *zero_word, void *stream);
int gpbo_launch_transpose_upper(const double *W, int64_t Np, double *U, hipStream_t st);
int gpbo_launch_argmax_finish(const double *part_val, const int64_t *part_idx, int64_t nparts,
                              const unsigned long long *nan_count, gpbo_result *result, hipStream_t st);
int64_t gpbo_posterior_workspace_bytes_split(int64_t Np, int64_t chunk, int64_t M, int split_max);
int gpbo_posterior_acq_f64_split(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                                 const double *ls_host, const double *U, const double *alpha, double prior_var,
                                 int32_t acq_kind, double p0, double p1, double diag_add, int64_t idx_offset,
                                 int64_t chunk, double *mu_out, double *sigma_out, double *acq_out, gpbo_result *result,
                                 void *work, int64_t work_bytes, gpbo_profile *prof, int split_max, int64_t n_prefix,
                                 void *stream);
#define GPBO_RESCORE_SPLIT_MAX 64
int gpbo_gemm_launch_tri(int transB, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                         int64_t strideA, const double *B, int64_t ldb, int64_t strideB, double beta, double *C,
                         int64_t ldc, int64_t strideC, int batch, int lower_only, int tri, hipStream_t st);
int gpbo_launch_split_finish(const double *ss_part, int S, int64_t ldk, const double *mu_part, int nsl, int64_t Mc,
                             double prior_var, int acq_kind, double p0, double p1, int64_t idx_base, double *mu_out,
                             double *sigma_out, double *acq_out, double *var_out, double *part_val, int64_t *part_idx,
                             unsigned long long *nan_count, hipStream_t st);
