// The 64 x 64 diagonal block of a blocked Cholesky: factor AND inverse factor in one elimination, in LDS.
// Shared by the single-matrix chain (factor.hip, potrf_diag_kernel) and the fused ARD likelihood kernel (ard.hip), so that
// both eliminate in the same order with the same arithmetic.  Replaces np.linalg.inv / np.linalg.det of one block
// (/root/reference/point_selector.py:89,117-118).
#pragma once
#include "gpbo_internal.h"

namespace gpbo_pd {

constexpr int NB = GPBO_NB;  // 64
// ---------------------------------------------------------------------------------------------
// Diagonal block: Cholesky of one 64x64 block AND the inverse of its factor, in one elimination.
// Waves 0..3 of the workgroup work, any further waves only join the barriers.  The eliminated matrix is the 128x64 stack M = [A; I] in LDS: carrying
// the identity rows through the same column operations leaves L^-T in them, so no separate triangular inversion
// is needed.  The 64 columns are taken 16 at a time:
//   (1) wave 0 eliminates the 16x16 diagonal sub-block and the identity rows under it in REGISTERS - lane = row,
//       16 values per lane, pivots and multipliers broadcast with v_readlane: no LDS round trip and no barrier
//       inside the 16 dependent column steps (the first version paid one barrier + LDS broadcast per column:
//       64 x 380 ns);
//   (2) the three non-zero 16x16 blocks of the panel are multiplied by the sub-block's L^-T on the matrix cores;
//   (3) the trailing 16x16 blocks are updated on the matrix cores - wave 0 takes the next diagonal sub-block
//       first and goes straight on to (1) for it while the other three waves do the rest.
// Two barriers per 16 columns.
// ---------------------------------------------------------------------------------------------
constexpr int PB = 16;      // inner block
constexpr int LDM = NB + 2; // row stride of M in LDS (doubles)

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rsqrt_refined(double p) {
    double y = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
    double e = fma(-(h * y), y, 0.5);
    y = fma(y, e, y);
    e = fma(-(h * y), y, 0.5);
    return fma(y, e, y);
}

// M [128 x LDM] in LDS: rows 0..63 = the block (lower triangle, zeros above), rows 64..127 = the identity.  On return
// rows 0..63 hold L (lower) and rows 64..127 hold L^-T (upper); every wave of the workgroup must call it (it contains
// workgroup barriers; the caller's data must be in place behind a barrier, and one has passed when it returns).
// Returns, in wave 0, the 1-based column of the first non-positive / non-finite pivot, or 0.
__device__ __forceinline__ int potrf_diag64_lds(double *M, int tid) {
    asm volatile("" : "+v"(tid));   // inside a caller's loop: the lane masks below are recomputed per call, not hoisted
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    int first_bad = 0;  // 1-based column of the first non-positive / non-finite pivot (wave 0)
    // (1): lanes 0..15 hold rows 16s.. of A, lanes 16..31 the identity rows under them (lanes 32..63 mirror
    // 0..31 and write nothing), columns 16s..16s+15
    auto eliminate = [&](int s) {
        const int l31 = lane & 31;
        const int prow = (l31 < PB) ? (PB * s + l31) : (NB + PB * s + (l31 - PB));
        double *row = M + prow * LDM + PB * s;
        double x[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) x[k] = row[k];
#pragma unroll
        for (int c = 0; c < PB; ++c) {
            const double piv = readlane_f64(x[c], c);
            const bool ok = (piv > 0.0) && (piv < 1.0e300);
            first_bad = (!ok && first_bad == 0) ? PB * s + c + 1 : first_bad;
            x[c] *= rsqrt_refined(piv);
#pragma unroll
            for (int k = c + 1; k < PB; ++k) x[k] = fma(-x[c], readlane_f64(x[c], k), x[k]);
        }
        if (lane < 2 * PB) {
#pragma unroll
            for (int k = 0; k < PB; ++k) row[k] = (lane < PB && k > lane) ? 0.0 : x[k];
        }
    };
    // (2): X <- X * T, X = 16 rows from R0 in column block s, T = L_ss^-T (rows 64+16s.. of M)
    auto panel = [&](int R0, int s) {
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < PB; kk += 4) {
            const double a = M[(R0 + l15) * LDM + PB * s + kk + l4];
            const double b = M[(NB + PB * s + kk + l4) * LDM + PB * s + l15];
            acc = mfma_f64_16x16x4(a, b, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) M[(R0 + l4 + 4 * r) * LDM + PB * s + l15] = acc[r];
    };
    // (3): M[R0.., block c] -= M[R0.., block s] * A[block c, block s]^T
    auto trail = [&](int R0, int c, int s) {
        d4_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = M[(R0 + l4 + 4 * r) * LDM + PB * c + l15];
#pragma unroll
        for (int kk = 0; kk < PB; kk += 4) {
            const double a = -M[(R0 + l15) * LDM + PB * s + kk + l4];
            const double b = M[(PB * c + l15) * LDM + PB * s + kk + l4];
            acc = mfma_f64_16x16x4(a, b, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) M[(R0 + l4 + 4 * r) * LDM + PB * c + l15] = acc[r];
    };

    constexpr int NS = NB / PB;  // 4
    if (w == 0) eliminate(0);
    gpbo_syncthreads();
    for (int s = 0; s < NS; ++s) {
        // (2) the non-zero panel blocks: A row blocks s+1..3 and identity-part row blocks 0..s-1 - always three
        if (w < NS - 1) {
            const int t = w;  // 0..2
            const int nA = NS - 1 - s;
            panel(t < nA ? PB * (s + 1 + t) : NB + PB * (t - nA), s);
        }
        gpbo_syncthreads();
        if (s == NS - 1) break;
        // (3) trailing blocks; wave 0: the next diagonal sub-block, then its elimination
        if (w == 0) {
            trail(PB * (s + 1), s + 1, s);
            eliminate(s + 1);
        } else if (w < NS) {
            int q = 0;
            for (int c = s + 1; c < NS; ++c) {
                for (int R = c; R < NS; ++R) {  // A part, lower block triangle
                    if (R == s + 1 && c == s + 1) continue;
                    if (q % 3 == w - 1) trail(PB * R, c, s);
                    ++q;
                }
                for (int m = 0; m <= s; ++m) {  // identity part: its rows 0..16(s+1)-1 are non-zero in block s
                    if (q % 3 == w - 1) trail(NB + PB * m, c, s);
                    ++q;
                }
            }
        }
        gpbo_syncthreads();
    }
    return first_bad;
}


}  // namespace gpbo_pd
