// Factorisation of K + sigma^2 I: blocked right-looking Cholesky, block-recursive
// inverse of the Cholesky factor, and alpha = K^-1 y.
//
// The reference inverts cov_meas with np.linalg.inv (/root/reference/point_selector.py:89) and uses
// the inverse twice (:90-91).  Here K = L L^T once per BO step; U = L^-T is formed explicitly so the
// per-candidate variance becomes a dense triangular product on the matrix cores
// (sigma_acq.hip) instead of a sequential triangular solve per candidate.
#include "gpbo_internal.h"

int gpbo_gemm_launch(int transB, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                     int64_t strideA, const double *B, int64_t ldb, int64_t strideB, double beta, double *C,
                     int64_t ldc, int64_t strideC, int batch, int lower_only, hipStream_t st);

namespace {

constexpr int NB = GPBO_NB;  // 64
constexpr int LDS_LD = NB + 1;

// ---------------------------------------------------------------------------------------------
// Diagonal block: Cholesky of one 64x64 block AND the inverse of its factor, in one elimination.
// One workgroup of 256 threads.  The eliminated matrix is the 128x64 stack [A; I]: carrying the identity
// rows through the same column operations leaves I * L^-T in them, i.e. inv(L)^T, so no separate
// triangular inversion (and none of its dependent LDS chains) is needed.
// Thread (ti, tj) keeps two 4x4 tiles in registers: rows 4ti.. of A and rows 4ti.. of the identity part,
// columns 4tj...  Per column c the 16 owning threads publish the UNSCALED column through LDS (double
// buffered, one barrier per column); everybody updates with a_ic * a_jc / piv, so only a reciprocal of the
// pivot is on the critical path - the square roots are taken once at the end for all 64 columns.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double rcp_refined(double p) {
    double y = __builtin_amdgcn_rcp(p);
    double e = fma(-p, y, 1.0);
    y = fma(y, e, y);
    e = fma(-p, y, 1.0);
    return fma(y, e, y);
}

__global__ __launch_bounds__(256) void potrf_diag_kernel(double *__restrict__ K, int64_t ld, int jb,
                                                         double *__restrict__ dinv, int32_t *__restrict__ info) {
    // colT[c][0..63]  = unscaled column c of the A part, colT[c][64..127] = of the identity part, as published
    // at step c.  Every column has its own slot, so the slot doubles as the record the final scaling reads
    // (no copy, no double buffering) and the loop body stays short: with one wave per SIMD nothing hides
    // instruction count, which is what bounded the earlier versions (~330 instructions per column).
    __shared__ double colT[NB][2 * NB];
    const int tid = threadIdx.x;
    double *Kd = K + ((int64_t)jb * NB) * ld + (int64_t)jb * NB;
    const int ti = tid >> 4, tj = tid & 15;
    double a[4][4], b[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 4 * ti + i, q = 4 * tj + j;
            a[i][j] = (q <= r) ? Kd[(int64_t)r * ld + q] : 0.0;
            b[i][j] = (q == r) ? 1.0 : 0.0;
        }
    int first_bad = 0;  // 1-based column of the first non-positive / non-finite pivot
    for (int cj = 0; cj < NB / 4; ++cj) {
        // columns to the right of the current one inside this thread's tile: all four if tj > cj, none if
        // tj < cj, (j > cc) on the diagonal tile column - applied as a mask on the broadcast column values
        const bool right = tj > cj, same = tj == cj;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = 4 * cj + cc;
            double *cb = colT[c];
            if (same) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cb[4 * ti + i] = a[i][cc];
                    cb[NB + 4 * ti + i] = b[i][cc];
                }
            }
            __syncthreads();
            const double piv = cb[c];
            const bool ok = (piv > 0.0) && (piv < 1.0e300);
            first_bad = (!ok && first_bad == 0) ? c + 1 : first_bad;
            const double rp = rcp_refined(piv);
            double ta[4], tb[4], aq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ta[i] = cb[4 * ti + i] * rp;
                tb[i] = cb[NB + 4 * ti + i] * rp;
                const bool upd = right || (same && i > cc);
                aq[i] = upd ? cb[4 * tj + i] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i][j] = fma(-ta[i], aq[j], a[i][j]);
                    b[i][j] = fma(-tb[i], aq[j], b[i][j]);
                }
        }
    }
    __syncthreads();
    if (tid == 0 && first_bad) atomicCAS(info, 0, jb * NB + first_bad);
    // scale: d_c = sqrt(piv_c);  L[r][c] = a_rc / d_c (r >= c);  inv(L)[r][c] = b_cr / d_r (c <= r)
    __shared__ double rs[NB];
    if (tid < NB) rs[tid] = 1.0 / sqrt(colT[tid][tid]);
    __syncthreads();
    double *dv = dinv + (int64_t)jb * NB * NB;
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        if (c <= r) Kd[(int64_t)r * ld + c] = colT[c][r] * rs[c];
        dv[e] = (c <= r) ? colT[r][NB + c] * rs[r] : 0.0;
    }
}

// W <- block-diagonal of dinv, zero elsewhere.   grid (Np/64, Np/64), block 256.
__global__ __launch_bounds__(256) void init_w_kernel(const double *__restrict__ dinv, double *__restrict__ W,
                                                     int64_t Np) {
    const int bi = blockIdx.y, bj = blockIdx.x;
    double *Wb = W + ((int64_t)bi * NB) * Np + (int64_t)bj * NB;
    const double *dv = dinv + (int64_t)bi * NB * NB;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        Wb[(int64_t)r * Np + c] = (bi == bj) ? dv[e] : 0.0;
    }
}

// U = W^T restricted to the upper triangle (W is lower triangular): 64x64 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_upper_kernel(const double *__restrict__ W, double *__restrict__ U,
                                                              int64_t Np) {
    __shared__ double tile[NB * LDS_LD];
    const int bi = blockIdx.y, bj = blockIdx.x;  // output tile (bi, bj) of U = input tile (bj, bi) of W
    double *Ub = U + ((int64_t)bi * NB) * Np + (int64_t)bj * NB;
    if (bj < bi) {  // strictly below the diagonal: zeros
        for (int e = threadIdx.x; e < NB * NB; e += 256) Ub[(int64_t)(e >> 6) * Np + (e & 63)] = 0.0;
        return;
    }
    const double *Wb = W + ((int64_t)bj * NB) * Np + (int64_t)bi * NB;
    for (int e = threadIdx.x; e < NB * NB; e += 256) tile[(e >> 6) * LDS_LD + (e & 63)] = Wb[(int64_t)(e >> 6) * Np + (e & 63)];
    __syncthreads();
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        double v = tile[c * LDS_LD + r];
        if (bi == bj && c < r) v = 0.0;
        Ub[(int64_t)r * Np + c] = v;
    }
}

// tmp_i = sum_{m<=i} U[m][i] y_m   (= (L^-1 y)_i).  Block = 16 columns (one 128-byte line per row of U) x 64 row
// groups: lane -> column lane&15, row group 4*wave + lane>>4 takes m = rg, rg+64, ...; fixed-order LDS reduction.
// (Np/16 workgroups: with 64 columns per workgroup a 4096-point problem kept only 64 of the 256 CUs busy.)
__global__ __launch_bounds__(1024) void utv_kernel(const double *__restrict__ U, const double *__restrict__ y,
                                                   int64_t N, int64_t Np, double *__restrict__ tmp) {
    __shared__ double part[64][17];
    const int lane = threadIdx.x & 63, col = lane & 15;
    const int rg = (threadIdx.x >> 6) * 4 + (lane >> 4);
    const int64_t i = (int64_t)blockIdx.x * 16 + col;
    double s = 0.0;
    const int64_t mend = (i < N ? i : N - 1);
    for (int64_t m = rg; m <= mend; m += 64) s = fma(U[m * Np + i], y[m], s);
    part[rg][col] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        double r = 0.0;
        for (int k = 0; k < 64; ++k) r += part[k][threadIdx.x];
        tmp[(int64_t)blockIdx.x * 16 + threadIdx.x] = r;
    }
}

// alpha_m = sum_{i>=m} U[m][i] tmp_i.  One wave per row m, lanes stride over i, fixed-order reduce.
__global__ __launch_bounds__(256) void uv_kernel(const double *__restrict__ U, const double *__restrict__ tmp,
                                                 int64_t N, int64_t Np, double *__restrict__ alpha) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= Np) return;
    double s = 0.0;
    if (m < N) {
        for (int64_t i = (m & ~63LL) + lane; i < Np; i += 64)
            if (i >= m) s = fma(U[m * Np + i], tmp[i], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) alpha[m] = (m < N) ? s : 0.0;
}

__global__ void zero_i32_kernel(int32_t *p) { *p = 0; }

}  // namespace

extern "C" int gpbo_potrf_f64(double *Kp, int64_t Np, double *dinv, int32_t *info, void *stream) {
    if (!Kp || !dinv || !info || Np < NB || Np % NB) return GPBO_ERR_ARG;
    hipStream_t st = gpbo_stream(stream);
    hipLaunchKernelGGL(zero_i32_kernel, dim3(1), dim3(1), 0, st, info);
    const int nb = (int)(Np / NB);
    for (int j = 0; j < nb; ++j) {
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, st, Kp, Np, j, dinv, info);
        GPBO_CHECK_LAUNCH();
        const int64_t rest = Np - (int64_t)(j + 1) * NB;
        if (rest <= 0) break;
        double *panel = Kp + (int64_t)(j + 1) * NB * Np + (int64_t)j * NB;  // [rest x 64]
        // panel <- panel * inv(L_jj)^T      (in place: each 64x64 tile is read whole before it is written)
        int rc = gpbo_gemm_launch(1, rest, NB, NB, 1.0, panel, Np, 0, dinv + (int64_t)j * NB * NB, NB, 0, 0.0, panel,
                                  Np, 0, 1, 0, st);
        if (rc != GPBO_OK) return rc;
        // trailing lower triangle -= panel * panel^T
        double *trail = Kp + (int64_t)(j + 1) * NB * Np + (int64_t)(j + 1) * NB;
        rc = gpbo_gemm_launch(1, rest, rest, NB, -1.0, panel, Np, 0, panel, Np, 0, 1.0, trail, Np, 0, 1, 1, st);
        if (rc != GPBO_OK) return rc;
    }
    return GPBO_OK;
}

int gpbo_launch_transpose_upper(const double *W, int64_t Np, double *U, hipStream_t st) {
    dim3 grid((unsigned)(Np / NB), (unsigned)(Np / NB));
    hipLaunchKernelGGL(transpose_upper_kernel, grid, dim3(256), 0, st, W, U, Np);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

extern "C" int gpbo_trtri_f64(const double *L, const double *dinv, int64_t Np, double *U, double *work,
                              void *stream) {
    if (!L || !dinv || !U || !work || Np < NB || Np % NB) return GPBO_ERR_ARG;
    hipStream_t st = gpbo_stream(stream);
    double *W = work;
    double *T = U;  // scratch until the final transpose
    dim3 g2((unsigned)(Np / NB), (unsigned)(Np / NB));
    hipLaunchKernelGGL(init_w_kernel, g2, dim3(256), 0, st, dinv, W, Np);
    GPBO_CHECK_LAUNCH();
    for (int64_t s = NB; s < Np; s <<= 1) {
        const int64_t full = Np / (2 * s);       // complete pairs (s, s)
        const int64_t rem = Np - full * 2 * s;   // what is left after them
        const int64_t stride = 2 * s * (Np + 1);
        if (full > 0) {
            // T21 = L21 * W11 ; W21 = -W22 * T21      (block (p): rows o+s.., cols o.., o = 2ps)
            int rc = gpbo_gemm_launch(0, s, s, s, 1.0, L + s * Np, Np, stride, W, Np, stride, 0.0, T + s * Np, Np,
                                      stride, (int)full, 0, st);
            if (rc != GPBO_OK) return rc;
            rc = gpbo_gemm_launch(0, s, s, s, -1.0, W + s * Np + s, Np, stride, T + s * Np, Np, stride, 0.0,
                                  W + s * Np, Np, stride, (int)full, 0, st);
            if (rc != GPBO_OK) return rc;
        }
        if (rem > s) {  // ragged last pair: W11 is s x s, W22 is m2 x m2 with m2 = rem - s < s
            const int64_t o = full * 2 * s, m2 = rem - s;
            const double *L21 = L + (o + s) * Np + o;
            int rc = gpbo_gemm_launch(0, m2, s, s, 1.0, L21, Np, 0, W + o * Np + o, Np, 0, 0.0, T + (o + s) * Np + o, Np,
                                      0, 1, 0, st);
            if (rc != GPBO_OK) return rc;
            rc = gpbo_gemm_launch(0, m2, s, m2, -1.0, W + (o + s) * Np + o + s, Np, 0, T + (o + s) * Np + o, Np, 0, 0.0,
                                  W + (o + s) * Np + o, Np, 0, 1, 0, st);
            if (rc != GPBO_OK) return rc;
        }
    }
    return gpbo_launch_transpose_upper(W, Np, U, st);
}

extern "C" int gpbo_alpha_f64(const double *U, const double *y, int64_t N, int64_t Np, double *tmp, double *alpha,
                              void *stream) {
    if (!U || !y || !tmp || !alpha || N < 1 || Np < N || Np % NB) return GPBO_ERR_ARG;
    hipStream_t st = gpbo_stream(stream);
    hipLaunchKernelGGL(utv_kernel, dim3((unsigned)(Np / 16)), dim3(1024), 0, st, U, y, N, Np, tmp);
    hipLaunchKernelGGL(uv_kernel, dim3((unsigned)((Np + 3) / 4)), dim3(256), 0, st, U, tmp, N, Np, alpha);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

extern "C" int64_t gpbo_factorise_workspace_bytes(int64_t Np) {
    // L [Np*Np] + trtri work [Np*Np] + dinv [Np*64] + tmp [Np]
    return (int64_t)sizeof(double) * (2 * Np * Np + Np * NB + Np);
}

extern "C" int gpbo_factorise_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_host,
                                  double jitter1, double jitter2, int64_t Np, double *Kp, double *U, double *alpha,
                                  int32_t *info, void *work, int64_t work_bytes, void *stream) {
    if (!X || !y || !Kp || !U || !alpha || !info || !work) return GPBO_ERR_ARG;
    if (Np != gpbo_padded_n(N)) return GPBO_ERR_ARG;
    if (work_bytes < gpbo_factorise_workspace_bytes(Np)) return GPBO_ERR_WORKSPACE;
    hipStream_t st = gpbo_stream(stream);
    double *L = reinterpret_cast<double *>(work);
    double *W = L + Np * Np;
    double *dinv = W + Np * Np;
    double *tmp = dinv + Np * NB;
    int rc = gpbo_kxx_f64(X, N, d, ls_host, jitter1, jitter2, Kp, Np, stream);
    if (rc != GPBO_OK) return rc;
    if (hipMemcpyAsync(L, Kp, sizeof(double) * Np * Np, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return GPBO_ERR_LAUNCH;
    rc = gpbo_potrf_f64(L, Np, dinv, info, stream);
    if (rc != GPBO_OK) return rc;
    rc = gpbo_trtri_f64(L, dinv, Np, U, W, stream);
    if (rc != GPBO_OK) return rc;
    return gpbo_alpha_f64(U, y, N, Np, tmp, alpha, stream);
}
