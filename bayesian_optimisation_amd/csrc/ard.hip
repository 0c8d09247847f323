// ARD length-scale grid search: -log marginal likelihood of every grid cell.
//
// Replaces PointSelector.tune_kernel / eval_log_marginal (/root/reference/point_selector.py:104-163):
//     rbf = kernel_rbf(X, X)            (1e-4 jitter only, :116,193)
//     nlml = 0.5 * (y^T inv(rbf) y + log(det(rbf)) + N log(2 pi))          (:117-119), stored as float32
// One workgroup per grid cell (N <= 176: the packed lower triangle fits in LDS).  The reference's inv + det become
// one Cholesky of the bordered matrix
//     [ K   y ]
//     [ y^T 0 ]
// carried N columns deep in LDS: the last row then holds (L^-1 y)^T, the last pivot is -y^T K^-1 y,
// and log det K = 2 sum log L_ii.  NumPy's det is sign * exp(logdet) (LAPACK getrf), so the reference's
// underflow of det to 0 (-> -inf cells for N >~ 100) is reproduced by evaluating log(exp(logdet)).
#include "gpbo_internal.h"

#include <cmath>
#include <limits>

namespace {

constexpr int ARD_MAX_N = 176;  // packed lower triangle of the (N+1) x (N+1) bordered matrix + X in LDS: 150 KB at d = 16

// PACKED = false: square image with row stride N + 2 (cheaper indexing, N <= 128); PACKED = true: lower triangle packed
// by rows (N <= 176).
template <bool PACKED>
__global__ __launch_bounds__(256) void nlml_grid_kernel(const double *__restrict__ X, const double *__restrict__ y,
                                                        int N, int d, const double *__restrict__ ls_cells,
                                                        double jitter, float *__restrict__ out) {
    extern __shared__ double lds[];
    // bordered matrix (N+1) x (N+1), lower triangle: entry (r, q <= r) at r (r + 1) / 2 + q when packed, else at
    // r (N + 2) + q (one more column than needed to break bank strides)
    const int ld = N + 2;
    auto tri = [ld](int r, int q) { return PACKED ? r * (r + 1) / 2 + q : r * ld + q; };
    double *a = lds;
    double *xs = a + (PACKED ? (N + 1) * (N + 2) / 2 : (N + 1) * ld);  // N x d
    double *il2 = xs + N * d;          // d
    double *diag = il2 + d;            // N   (L_ii)
    const int tid = threadIdx.x;
    const double *ls = ls_cells + (int64_t)blockIdx.x * d;

    for (int e = tid; e < N * d; e += 256) xs[e] = X[e];
    if (tid < d) il2[tid] = 1.0 / (ls[tid] * ls[tid]);
    __syncthreads();
    const int n1 = N + 1;
    for (int e = tid; e < n1 * n1; e += 256) {
        const int r = e / n1, q = e - r * n1;
        if (q > r) continue;
        double v = 0.0;
        {
            if (r < N) {
                double acc = 0.0;
                for (int k = 0; k < d; ++k) {
                    const double diff = xs[r * d + k] - xs[q * d + k];
                    acc = fma(diff * diff, il2[k], acc);
                }
                v = exp(-0.5 * acc);
                if (r == q) v += jitter;
            } else if (q < N) {
                v = y[q];
            }
        }
        a[tri(r, q)] = v;
    }
    for (int c = 0; c < N; ++c) {
        __syncthreads();
        const double piv = a[tri(c, c)];
        const double dd = sqrt(piv);   // NaN for a negative pivot: propagates to the cell's value
        const double inv = 1.0 / dd;
        if (tid == 0) diag[c] = dd;
        // trailing update on rows r > c, columns c < q <= r of the bordered matrix
        const int m = N - c;           // rows c+1 .. N
        for (int e = tid; e < m * m; e += 256) {
            const int r = c + 1 + e / m, q = c + 1 + (e - (e / m) * m);
            if (q <= r) {
                const double lr = a[tri(r, c)] * inv;
                const double lq = a[tri(q, c)] * inv;
                a[tri(r, q)] = fma(-lr, lq, a[tri(r, q)]);
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        double logdet = 0.0;
        for (int c = 0; c < N; ++c) logdet += log(diag[c]);
        logdet *= 2.0;
        const double quad = -a[tri(N, N)];
        const double logdet_ref = log(exp(logdet));  // the reference takes log of an underflowing det
        const double nlml = 0.5 * (quad + logdet_ref + (double)N * 1.8378770664093453);  // log(2 pi)
        out[blockIdx.x] = (float)nlml;
    }
}

// One grid cell from an existing factorisation (any N): log det K = -2 sum log U_ii (U = L^-T), y^T K^-1 y = y . alpha.
__global__ __launch_bounds__(256) void nlml_cell_kernel(const double *__restrict__ U, const double *__restrict__ alpha,
                                                        const double *__restrict__ y, int64_t N, int64_t Np,
                                                        const int32_t *__restrict__ info, float *__restrict__ out) {
    __shared__ double s_ld[256], s_q[256];
    const int tid = threadIdx.x;
    double ld = 0.0, q = 0.0;
    for (int64_t i = tid; i < N; i += 256) {
        ld -= log(U[i * Np + i]);
        q = fma(y[i], alpha[i], q);
    }
    s_ld[tid] = ld;
    s_q[tid] = q;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) { s_ld[tid] += s_ld[tid + off]; s_q[tid] += s_q[tid + off]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double logdet = 2.0 * s_ld[0];
        const double logdet_ref = log(exp(logdet));  // the reference takes log of a det that under/overflows
        double nlml = 0.5 * (s_q[0] + logdet_ref + (double)N * 1.8378770664093453);
        if (*info != 0) nlml = __builtin_nan("");   // not positive definite: the reference's log(det < 0) is NaN
        *out = (float)nlml;
    }
}

// ---- any N: all cells of a batch through one blocked Cholesky (factor.hip, gpbo_potrf_batched) ----------------------
// Cell g owns a bordered matrix [Ne x Ne], Ne = 64 ceil(N/64) + 64:
//     rows/cols < N            k(x_i, x_j) with the cell's length scales, + jitter on the diagonal
//     rows N .. Nf-1           identity (padding of the factorised part, Nf = 64 ceil(N/64))
//     row  Nf                  y^T (columns < N), zero elsewhere: after Nf columns of the factorisation it holds
//                              (L^-1 y)^T and entry (Nf, Nf) holds -y^T K^-1 y - the bordered-matrix identity the
//                              in-LDS kernel above uses, here carried by the panel solve / trailing update GEMMs
//     rows > Nf                identity, never factorised.
// grid (Ne/256 up, Ne/8, cells), block 256: thread = column, 8 rows per workgroup (as kxx_kernel).
template <int D>
__global__ __launch_bounds__(256) void nlml_build_kernel(const double *__restrict__ X, const double *__restrict__ y, int N,
                                                         int Nf, int Ne, const double *__restrict__ ls_cells,
                                                         double jitter, double *__restrict__ Ab,
                                                         int32_t *__restrict__ info) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int64_t g = blockIdx.z;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) info[g] = 0;
    if (j >= Ne) return;
    const double *ls = ls_cells + g * D;
    double il2[D], xj[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        il2[k] = 1.0 / (ls[k] * ls[k]);
        xj[k] = (j < N) ? X[(int64_t)j * D + k] : 0.0;
    }
    double *A = Ab + g * (int64_t)Ne * Ne;
    const int i0 = blockIdx.y * 8;
    for (int i = i0; i < i0 + 8; ++i) {
        double v;
        if (i < N && j < N) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double diff = xj[k] - X[(int64_t)i * D + k];
                acc = fma(diff * diff, il2[k], acc);
            }
            v = exp(-0.5 * acc);
            if (i == j) v += jitter;
        } else if (i == Nf) {
            v = (j < N) ? y[j] : 0.0;
        } else {
            v = (i == j) ? 1.0 : 0.0;
        }
        A[(int64_t)i * Ne + j] = v;
    }
}

// one workgroup per cell: log det K = 2 sum log L_ii, y^T K^-1 y = -A[Nf][Nf]; NaN when a pivot failed
__global__ __launch_bounds__(256) void nlml_finish_kernel(const double *__restrict__ Ab, int N, int Nf, int Ne,
                                                          const int32_t *__restrict__ info, float *__restrict__ out) {
    __shared__ double s_ld[256];
    const int tid = threadIdx.x;
    const double *A = Ab + (int64_t)blockIdx.x * Ne * Ne;
    double ld = 0.0;
    for (int i = tid; i < N; i += 256) ld += log(A[(int64_t)i * Ne + i]);
    s_ld[tid] = ld;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) s_ld[tid] += s_ld[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        const double logdet = 2.0 * s_ld[0];
        const double quad = -A[(int64_t)Nf * Ne + Nf];
        const double logdet_ref = log(exp(logdet));  // the reference takes log of a det that under/overflows
        double nlml = 0.5 * (quad + logdet_ref + (double)N * 1.8378770664093453);
        if (info[blockIdx.x] != 0) nlml = __builtin_nan("");
        out[blockIdx.x] = (float)nlml;
    }
}

inline int64_t nlml_batch_cells(int64_t Ne, int64_t G) {
    // matrices of one sub-batch stay under 8 GiB (2,500 cells at N = 512 are 6.6 GB: one batch)
    int64_t cap = (int64_t)(8ll << 30) / (Ne * Ne * 8);
    if (cap < 1) cap = 1;
    if (cap > 32768) cap = 32768;
    return G < cap ? G : cap;
}

}  // namespace

extern "C" int64_t gpbo_nlml_grid_batched_workspace_bytes(int64_t N, int64_t G) {
    if (N < 1 || G < 1 || N > (1 << 15)) return GPBO_ERR_ARG;
    const int64_t Nf = (N + GPBO_NB - 1) / GPBO_NB * GPBO_NB, Ne = Nf + GPBO_NB;
    const int64_t B = nlml_batch_cells(Ne, G);
    return B * (Ne * Ne * 8 + (Nf / GPBO_NB) * GPBO_NB * GPBO_NB * 8 + 256) + 256;
}

extern "C" int gpbo_nlml_grid_batched_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells,
                                          int64_t G, double jitter, float *out, void *work, int64_t work_bytes,
                                          void *stream) {
    if (!X || !y || !ls_cells || !out || !work || N < 1 || d < 1 || d > GPBO_MAX_D || G < 1) return GPBO_ERR_ARG;
    const int64_t need = gpbo_nlml_grid_batched_workspace_bytes(N, G);
    if (need < 0) return GPBO_ERR_ARG;
    if (work_bytes < need || ((uintptr_t)work & 255)) return GPBO_ERR_WORKSPACE;
    const int64_t Nf = (N + GPBO_NB - 1) / GPBO_NB * GPBO_NB, Ne = Nf + GPBO_NB;
    const int64_t B = nlml_batch_cells(Ne, G);
    const int nbf = (int)(Nf / GPBO_NB);
    char *w = reinterpret_cast<char *>(work);
    double *Ab = reinterpret_cast<double *>(w);
    double *dinv = Ab + B * Ne * Ne;
    int32_t *info = reinterpret_cast<int32_t *>(dinv + B * nbf * GPBO_NB * GPBO_NB);
    hipStream_t st = gpbo_stream(stream);
    for (int64_t g0 = 0; g0 < G; g0 += B) {
        const int64_t nb = (G - g0 < B) ? (G - g0) : B;
        dim3 grid((unsigned)((Ne + 255) / 256), (unsigned)(Ne / 8), (unsigned)nb);
#define CALL(DD)                                                                                                       \
    hipLaunchKernelGGL(nlml_build_kernel<DD>, grid, dim3(256), 0, st, X, y, (int)N, (int)Nf, (int)Ne, ls_cells + g0 * d,   \
                       jitter, Ab, info)
        switch (d) {
            case 1: CALL(1); break;   case 2: CALL(2); break;   case 3: CALL(3); break;   case 4: CALL(4); break;
            case 5: CALL(5); break;   case 6: CALL(6); break;   case 7: CALL(7); break;   case 8: CALL(8); break;
            case 9: CALL(9); break;   case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break;
            case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; case 16: CALL(16); break;
            default: return GPBO_ERR_ARG;
        }
#undef CALL
        GPBO_CHECK_LAUNCH();
        int rc = gpbo_potrf_batched(Ab, Ne, nbf, (int)nb, dinv, info, st);
        if (rc != GPBO_OK) return rc;
        hipLaunchKernelGGL(nlml_finish_kernel, dim3((unsigned)nb), dim3(256), 0, st, Ab, (int)N, (int)Nf, (int)Ne, info,
                           out + g0);
        GPBO_CHECK_LAUNCH();
    }
    return GPBO_OK;
}

extern "C" int gpbo_nlml_cell_f64(const double *U, const double *alpha, const double *y, int64_t N, int64_t Np,
                                  const int32_t *info, float *out, void *stream) {
    if (!U || !alpha || !y || !info || !out || N < 1 || Np < N) return GPBO_ERR_ARG;
    hipLaunchKernelGGL(nlml_cell_kernel, dim3(1), dim3(256), 0, gpbo_stream(stream), U, alpha, y, N, Np, info, out);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

extern "C" int gpbo_nlml_grid_max_n(void) { return ARD_MAX_N; }

extern "C" int gpbo_nlml_grid_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells,
                                  int64_t G, double jitter, float *out, void *stream) {
    if (!X || !y || !ls_cells || !out || N < 1 || N > ARD_MAX_N || d < 1 || d > GPBO_MAX_D || G < 1 ||
        G > (1 << 30))
        return GPBO_ERR_ARG;
    const bool packed = N > 128;
    const size_t mat = packed ? (size_t)(N + 1) * (N + 2) / 2 : (size_t)(N + 1) * (N + 2);
    const size_t lds_bytes = sizeof(double) * (mat + (size_t)N * d + d + N);
    const void *fn = packed ? reinterpret_cast<const void *>(nlml_grid_kernel<true>)
                            : reinterpret_cast<const void *>(nlml_grid_kernel<false>);
    if (lds_bytes > 64 * 1024) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return GPBO_ERR_LAUNCH;
    }
    if (packed)
        hipLaunchKernelGGL(nlml_grid_kernel<true>, dim3((unsigned)G), dim3(256), lds_bytes, gpbo_stream(stream), X, y,
                           (int)N, (int)d, ls_cells, jitter, out);
    else
        hipLaunchKernelGGL(nlml_grid_kernel<false>, dim3((unsigned)G), dim3(256), lds_bytes, gpbo_stream(stream), X, y,
                           (int)N, (int)d, ls_cells, jitter, out);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}
