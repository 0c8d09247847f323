// Appending one observation to an existing factorisation in O(N^2)  (SURVEY.md §8f rank 4).
//
// Consecutive BO iterations of the reference differ by one observed row
// (/root/reference/select_parameters.py:163,299 append it; :142,265 reload everything and
// /root/reference/point_selector.py:89 inverts the full matrix again).  With the factors this library keeps,
//     K' = [K k; k^T kappa],  L' = [L 0; l^T lambda],  l = L^-1 k = U^T k,  lambda^2 = kappa - l.l
//     U' = L'^-T = [U  -U l / lambda; 0  1 / lambda]
// so only column N of U is new; alpha' = U' (U'^T y') is recomputed in full (two matrix-vector products,
// same kernels and summation order as gpbo_alpha_f64, so nothing accumulates over repeated appends).
// K entries use the arithmetic of kxx_kernel (kernel_build.hip) so that an appended row of K is bit-identical
// to the row a fresh gpbo_kxx_f64 would build.
#include "gpbo_internal.h"

#include <cmath>

namespace {

struct LsInv2 {
    double il2[GPBO_MAX_D];  // 1 / ls_k^2, host fp64 (as kernel_build.hip's LsArgs::il2)
};

// kvec_i = k(x_i, x_new) for i < N, 0 on the padding; row N of X and y receive the new observation.
// grid ceil(Np/256), block 256.
__global__ __launch_bounds__(256) void append_kvec_kernel(double *__restrict__ X, double *__restrict__ y, int N, int d,
                                                          LsInv2 ls, const double *__restrict__ x_new,
                                                          const double *__restrict__ y_new, int Np,
                                                          double *__restrict__ kvec) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Np) return;
    double v = 0.0;
    if (i < N) {
        double acc = 0.0;
        for (int k = 0; k < d; ++k) {
            const double diff = X[(int64_t)i * d + k] - x_new[k];
            acc = fma(diff * diff, ls.il2[k], acc);
        }
        v = exp(-0.5 * acc);
    }
    kvec[i] = v;
    if (i == N) {
        for (int k = 0; k < d; ++k) X[(int64_t)N * d + k] = x_new[k];
        y[N] = y_new[0];
    }
}

// lambda^2 = kappa - sum_j l_j^2 in a fixed order; scal[0] = 1/lambda, info = 0 or N+1 (1-based failing pivot).
// One block of 1024 threads.
__global__ __launch_bounds__(1024) void append_pivot_kernel(const double *__restrict__ l, int N, double kappa,
                                                            double *__restrict__ scal, int32_t *__restrict__ info) {
    __shared__ double part[1024];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int j = tid; j < N; j += 1024) s = fma(l[j], l[j], s);
    part[tid] = s;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) part[tid] += part[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        const double lam2 = kappa - part[0];
        const bool ok = lam2 > 0.0 && lam2 <= kappa;  // also false for NaN
        scal[0] = ok ? 1.0 / sqrt(lam2) : 0.0;
        *info = ok ? 0 : N + 1;
    }
}

// Column N of U (and row/column N of K when kept): U[i][N] = -c_i / lambda, U[N][N] = 1 / lambda.
// Nothing is written when the pivot failed.  grid ceil(Np/256), block 256.
__global__ __launch_bounds__(256) void append_column_kernel(const double *__restrict__ c, const double *__restrict__ kvec,
                                                            const double *__restrict__ scal,
                                                            const int32_t *__restrict__ info, int N, int64_t Np,
                                                            double kappa, double *__restrict__ U,
                                                            double *__restrict__ Kp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > N || *info != 0) return;
    const double inv = scal[0];
    U[(int64_t)i * Np + N] = (i < N) ? -c[i] * inv : inv;
    if (Kp) {
        const double kv = (i < N) ? kvec[i] : kappa;
        Kp[(int64_t)i * Np + N] = kv;
        Kp[(int64_t)N * Np + i] = kv;
    }
}

// alpha is committed only when the pivot was accepted: a failed append leaves the surrogate exactly as it was
// (U untouched, alpha untouched), so a caller that catches the error and keeps scoring gets the old posterior.
__global__ __launch_bounds__(256) void append_commit_alpha_kernel(const double *__restrict__ src,
                                                                  const int32_t *__restrict__ info, int64_t Np,
                                                                  double *__restrict__ alpha) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < Np && *info == 0) alpha[i] = src[i];
}

}  // namespace

extern "C" int64_t gpbo_append_workspace_bytes(int64_t Np) {
    if (Np < 1) return -1;
    return (int64_t)sizeof(double) * (3 * Np + 8);  // kvec, l, c, scalars
}

extern "C" int gpbo_append_f64(double *X, double *y, int64_t N, int32_t d, const double *ls_host, double jitter1,
                               double jitter2, int64_t Np, const double *x_new, const double *y_new, double *Kp,
                               double *U, double *alpha, int32_t *info, void *work, int64_t work_bytes,
                               void *stream) {
    if (!X || !y || !ls_host || !x_new || !y_new || !U || !alpha || !info || !work) return GPBO_ERR_ARG;
    if (N < 1 || d < 1 || d > GPBO_MAX_D || Np % GPBO_NB || N + 1 > Np || Np > (1 << 30)) return GPBO_ERR_ARG;
    if (work_bytes < gpbo_append_workspace_bytes(Np)) return GPBO_ERR_WORKSPACE;
    LsInv2 ls;
    for (int k = 0; k < GPBO_MAX_D; ++k) ls.il2[k] = 0.0;
    for (int k = 0; k < d; ++k) {
        if (!(ls_host[k] > 0.0)) return GPBO_ERR_ARG;
        ls.il2[k] = 1.0 / (ls_host[k] * ls_host[k]);
    }
    hipStream_t st = gpbo_stream(stream);
    double *kvec = reinterpret_cast<double *>(work);
    double *l = kvec + Np;
    double *c = l + Np;
    double *scal = c + Np;
    const double kappa = (1.0 + jitter1) + jitter2;  // diagonal of K as kxx_kernel rounds it (exp(0) = 1)
    const unsigned nblk = (unsigned)((Np + 255) / 256);
    hipLaunchKernelGGL(append_kvec_kernel, dim3(nblk), dim3(256), 0, st, X, y, (int)N, (int)d, ls, x_new, y_new,
                       (int)Np, kvec);
    GPBO_CHECK_LAUNCH();
    int rc = gpbo_alpha_f64(U, kvec, N, Np, l, c, stream);  // l = U^T k,  c = U l
    if (rc != GPBO_OK) return rc;
    hipLaunchKernelGGL(append_pivot_kernel, dim3(1), dim3(1024), 0, st, l, (int)N, kappa, scal, info);
    hipLaunchKernelGGL(append_column_kernel, dim3(nblk), dim3(256), 0, st, c, kvec, scal, info, (int)N, Np, kappa, U,
                       Kp);
    GPBO_CHECK_LAUNCH();
    // alpha of the N+1 observations goes to the workspace first (c is free again) and is committed under info == 0
    rc = gpbo_alpha_f64(U, y, N + 1, Np, l, c, stream);
    if (rc != GPBO_OK) return rc;
    hipLaunchKernelGGL(append_commit_alpha_kernel, dim3(nblk), dim3(256), 0, st, c, info, Np, alpha);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}
