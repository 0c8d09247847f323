// Strided-batched fp64 GEMM on the gfx950 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Serves the factorisation only: the Cholesky panel solve and trailing
// SYRK/GEMM update, and the block-recursive triangular inverse.  None of this exists in the
// reference, which calls np.linalg.inv (/root/reference/point_selector.py:89).
//
// C_b = alpha * A_b * op(B_b) + beta * C_b, all row-major.
//   workgroup = 256 threads = 4 waves in a 2x2 arrangement, 64x64 tile of C (32x32 for small products), BK = 16;
//   wave tile 32x32 = 2x2 MFMA tiles (16x16 = one tile in the small variant).
// LDS images are chosen so that staging is a straight, coalesced copy of the global layout:
//   A  [M x K] row-major -> As[m][k], row stride 17 doubles (ds_read_b64 conflict-free: lanes
//                           l and l+16 of a 32-lane group read k and k+1 of 16 different rows)
//   B  [K x N] row-major -> Bs[k][n], row stride T+16 doubles (lanes 16-31 land 32 banks away)
//   B' [N x K] row-major -> As-style image [n][k], stride 17.
#include "gpbo_internal.h"

namespace {

constexpr int BK = 16;
constexpr int LDA_S = 17;  // As[m][k] / Bt[n][k] row stride (doubles)

// T = tile edge (64 or 32).  The 32 x 32 variant serves the small products of the factorisation (N <= 1024: a
// 448 x 448 trailing update is 28 tiles of 64 x 64 - 28 of 256 CUs - and every k tile then costs 16 dependent MFMAs
// per wave; with 32 x 32 tiles the same update runs on 105 workgroups with 4 MFMAs per k tile).
// tri: 1 = B [K x N] is lower triangular (rows k < n are zero): the k loop of column tile bx starts at bx T;
//      2 = A [M x K] is lower triangular (columns k > m are zero): the k loop of row tile by ends at (by + 1) T
// (the two products of each level of the triangular inverse; N = 4096: factorisation 3.42 -> 3.17 ms).
// Why 38-54 TFLOP/s: the 64 x 64 tile moves 16 KB per 16-deep k tile for 2 x 64 x 64 x 16 flop = 8 flop/B, and at the
// ~12 B/clk/CU these kernels get out of L2 (the rate ozaki.hip measures for its operands too) that is ~52 TFLOP/s.
// 128 x 128 tiles double the intensity; both forms were built and measured (tools/bench_gemm.py) and are not kept:
//  - 4 waves, wave tile 64 x 64 (210 registers, two waves per SIMD): slower on every shape (8192^2 x 256 lower-triangle
//    update 29.5 against 39.3 TFLOP/s, 4096^3 49.8 against 54.2);
//  - 8 waves, wave tile 32 x 64 (122 registers): 68 TFLOP/s at 4096^3, but the factorisation is made of K <= 256 updates,
//    where a quarter as many tiles leaves compute units idle (4096^2 x 128: 28 against 39 TFLOP/s; whole factorisation at
//    N = 8192 12.6 against 12.3 ms) and the batched ARD products got 18x slower.
template <int TRANSB, int T>
__global__ __launch_bounds__(256) void gemm_f64_kernel(int64_t M, int64_t N, int64_t K, double alpha,
                                                        const double *__restrict__ A, int64_t lda, int64_t strideA,
                                                        const double *__restrict__ B, int64_t ldb, int64_t strideB,
                                                        double beta, double *__restrict__ C, int64_t ldc,
                                                        int64_t strideC, int lower_only, int tri) {
    constexpr int TM = T, TN = T;
    constexpr int FM = T / 32;            // 16 x 16 MFMA tiles per wave in each direction (wave tile T/2 x T/2)
    constexpr int LDB_S = T + 16;         // Bs[k][n] row stride (doubles): lanes 16-31 land 32 banks away
    constexpr int STAGERS = T * BK / 4;   // threads that stage one operand tile, 4 doubles each (256 or 128)
    const int bx = blockIdx.x, by = blockIdx.y;  // bx: column tile, by: row tile
    if (lower_only && (bx * T) / 64 > (by * T) / 64) return;  // 64 x 64 granularity in both variants
    __shared__ double As[TM * LDA_S];
    __shared__ double Bs[(TRANSB ? TN * LDA_S : BK * LDB_S)];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    A += (int64_t)blockIdx.z * strideA + (int64_t)by * TM * lda;
    B += (int64_t)blockIdx.z * strideB;
    C += (int64_t)blockIdx.z * strideC + (int64_t)by * TM * ldc + (int64_t)bx * TN;
    if (TRANSB) B += (int64_t)bx * TN * ldb; else B += (int64_t)bx * TN;

    d4_t acc[FM][FM];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};

    // staging indices: A-style tile = T rows x 16 k: thread -> row tid/4, 4 doubles at k = (tid%4)*4
    const int ar = tid >> 2, ak = (tid & 3) * 4;
    // B (no-trans) tile = 16 k-rows x T cols: thread -> k-row tid/(T/4), 4 doubles at col (tid%(T/4))*4
    const int bk = tid / (T / 4), bn = (tid % (T / 4)) * 4;
    const bool stager = tid < STAGERS;

    // register prefetch: the global loads of k tile t+1 are in flight while tile t is multiplied (the panel
    // products of the factorisation have K = 64: without this every one of their four k tiles exposes a full
    // global-memory round trip)
    d2_t a0 = {0.0, 0.0}, a1 = a0, b0 = a0, b1 = a0;
    auto gload = [&](int64_t k0) {
        if (!stager) return;
        const d2_t *ap = reinterpret_cast<const d2_t *>(A + (int64_t)ar * lda + k0 + ak);
        a0 = ap[0]; a1 = ap[1];
        if (TRANSB) {
            const d2_t *bp = reinterpret_cast<const d2_t *>(B + (int64_t)ar * ldb + k0 + ak);
            b0 = bp[0]; b1 = bp[1];
        } else {
            const d2_t *bp = reinterpret_cast<const d2_t *>(B + (k0 + bk) * ldb + bn);
            b0 = bp[0]; b1 = bp[1];
        }
    };
    // k range of this tile (triangular operands of the inverse: the zero part is never read)
    int64_t kbeg = 0, kend = K;
    if (tri == 1) kbeg = (int64_t)bx * T;
    if (tri == 2 && (int64_t)(by + 1) * T < K) kend = (int64_t)(by + 1) * T;
    gload(kbeg);
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();  // previous tile fully consumed
        if (stager) {
            As[ar * LDA_S + ak + 0] = a0.x; As[ar * LDA_S + ak + 1] = a0.y;
            As[ar * LDA_S + ak + 2] = a1.x; As[ar * LDA_S + ak + 3] = a1.y;
            if (TRANSB) {
                Bs[ar * LDA_S + ak + 0] = b0.x; Bs[ar * LDA_S + ak + 1] = b0.y;
                Bs[ar * LDA_S + ak + 2] = b1.x; Bs[ar * LDA_S + ak + 3] = b1.y;
            } else {
                *reinterpret_cast<d2_t *>(&Bs[bk * LDB_S + bn]) = b0;
                *reinterpret_cast<d2_t *>(&Bs[bk * LDB_S + bn + 2]) = b1;
            }
        }
        __syncthreads();
        if (k0 + BK < kend) gload(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            double af[FM], bf[FM];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = As[(wr * (T / 2) + i * 16 + l15) * LDA_S + kk + l4];
#pragma unroll
            for (int j = 0; j < FM; ++j)
                bf[j] = TRANSB ? Bs[(wc * (T / 2) + j * 16 + l15) * LDA_S + kk + l4]
                               : Bs[(kk + l4) * LDB_S + wc * (T / 2) + j * 16 + l15];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = mfma_f64_16x16x4(af[i], bf[j], acc[i][j]);
        }
    }

#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wr * (T / 2) + i * 16 + l4 + 4 * r;
                const int col = wc * (T / 2) + j * 16 + l15;
                double *cp = C + (int64_t)row * ldc + col;
                double v = alpha * acc[i][j][r];
                if (beta != 0.0) v = fma(beta, *cp, v);
                *cp = v;
            }
}

}  // namespace

int gpbo_gemm_launch(int transB, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                     int64_t strideA, const double *B, int64_t ldb, int64_t strideB, double beta, double *C,
                     int64_t ldc, int64_t strideC, int batch, int lower_only, hipStream_t st) {
    return gpbo_gemm_launch_tri(transB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch,
                                lower_only, 0, st);
}

int gpbo_gemm_launch_tri(int transB, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                         int64_t strideA, const double *B, int64_t ldb, int64_t strideB, double beta, double *C,
                         int64_t ldc, int64_t strideC, int batch, int lower_only, int tri, hipStream_t st) {
    if (M <= 0 || N <= 0 || batch <= 0) return GPBO_OK;
    if (tri < 0 || tri > 2 || (tri == 1 && (transB || K != N)) || (tri == 2 && K != M)) return GPBO_ERR_ARG;
    if (!A || !B || !C || M % 64 || N % 64 || K % BK || K <= 0 || (lda & 1) || (ldb & 1)) return GPBO_ERR_ARG;
    if (((uintptr_t)A | (uintptr_t)B) & 15) return GPBO_ERR_ARG;
    if (N / 32 > 65535 || M / 32 > 65535 || batch > 65535) return GPBO_ERR_ARG;
    // up to 2048 tiles of 64 x 64 (8 per CU) the 32 x 32 variant wins: four times the workgroups, a quarter of the
    // dependent MFMAs per k tile.  Measured, whole factorisation in ms with the switch at 128 / 512 / 2048 / 8192 tiles:
    // N = 2048: 1.28 / 1.17 / 1.17 / 1.18;  N = 4096: 3.55 / 3.48 / 3.41 / 3.41;  N = 8192: 15.1 / - / 14.7 / 16.1
    int64_t tiles64 = (M / 64) * (N / 64) * batch;
    if (lower_only) tiles64 = (M / 64) * (M / 64 + 1) / 2 * batch;
    // (not when C aliases an operand: the in-place panel solve relies on one workgroup owning a whole 64-row tile,
    //  which it reads completely before it writes)
    const bool small = tiles64 < 2048 && C != A && C != B;
    const int T = small ? 32 : 64;
    dim3 grid((unsigned)(N / T), (unsigned)(M / T), (unsigned)batch);
#define GPBO_GEMM_LAUNCH(TB, TT)                                                                                        \
    hipLaunchKernelGGL((gemm_f64_kernel<TB, TT>), grid, dim3(256), 0, st, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, \
                       beta, C, ldc, strideC, lower_only, tri)
    if (transB) {
        if (small) GPBO_GEMM_LAUNCH(1, 32); else GPBO_GEMM_LAUNCH(1, 64);
    } else {
        if (small) GPBO_GEMM_LAUNCH(0, 32); else GPBO_GEMM_LAUNCH(0, 64);
    }
#undef GPBO_GEMM_LAUNCH
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

extern "C" int gpbo_gemm_f64(int32_t transB, int64_t M, int64_t N, int64_t K, double alpha, const double *A,
                             int64_t lda, int64_t strideA, const double *B, int64_t ldb, int64_t strideB,
                             double beta, double *C, int64_t ldc, int64_t strideC, int32_t batch,
                             int32_t lower_only, void *stream) {
    return gpbo_gemm_launch(transB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, beta, C, ldc, strideC, batch,
                            lower_only, gpbo_stream(stream));
}
