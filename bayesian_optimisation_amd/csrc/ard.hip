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
#include "exp_neg.h"
#include "potrf_diag64.h"
#include <atomic>

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
    gpbo_syncthreads();
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
        gpbo_syncthreads();
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
    gpbo_syncthreads();
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
    gpbo_syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) { s_ld[tid] += s_ld[tid + off]; s_q[tid] += s_q[tid + off]; }
        gpbo_syncthreads();
    }
    if (tid == 0) {
        const double logdet = 2.0 * s_ld[0];
        const double logdet_ref = log(exp(logdet));  // the reference takes log of a det that under/overflows
        double nlml = 0.5 * (s_q[0] + logdet_ref + (double)N * 1.8378770664093453);
        if (*info != 0) nlml = __builtin_nan("");   // not positive definite: the reference's log(det < 0) is NaN
        *out = (float)nlml;
    }
}

// ---- any N: one workgroup per grid cell, the whole factorisation in ONE launch (round 5) ------------------------------
// Left-looking blocked Cholesky of the cell's K, 64 columns (one PANEL) at a time, K's entries generated on the fly:
//     C' = K[rows, panel]^T - L[panel rows, :J0] L[rows, :J0]^T      matrix cores, operands straight from memory
//     D  = C' of the panel's own 64 rows -> L_jj, W = inv(L_jj)       potrf_diag64_lds (LDS)
//     L[rows below, panel]^T = W C'                                   matrix cores, A operand from LDS, B = the accumulators
// y rides along as one more row (the bordered matrix of nlml_grid_kernel): its row of L is z^T = (L^-1 y)^T, and
// y^T K^-1 y = |z|^2.  log det K = 2 sum log L_ii.
//
// Everything is computed TRANSPOSED (C'[panel column][row]) because of how v_mfma_f64_16x16x4_f64 lays its operands out:
// register r of lane l of a result tile is C'[(l >> 4) + 4r][l & 15], which is exactly what lane l must supply as the B
// operand of k-group r - so the accumulators of the first product ARE the B operands of the second, and the second product's
// accumulators ARE the fragments later panels load as operands.  Nothing is transposed through LDS, and L lives in memory
// in FRAGMENT ORDER:   frag(rt, kg)[lane] = L[16 rt + (lane & 15)][4 kg + (lane >> 4)], two k-groups interleaved per lane
// (one 16-byte load / store per lane, 1 KiB contiguous per wave instruction).
// The workgroup is persistent (cells g = blockIdx.x, + gridDim.x, ...; as many workgroups as are resident at once: two per
// CU) and owns one scratch slot of (Nf + 16) x Nf doubles, written once and read (N / 64) / 3 times on average per cell -
// the matrices of the launch-chain version of rounds 2-4 (up to 8 GiB per sub-batch, swept once per panel by three
// launches: 11.4 / 61 ms for 2,500 cells at N = 512 / 1024) are gone.  Measured (one MI355X, 2,500 cells, d = 8):
// N = 176 / 512 / 1024: 0.63 / 3.9 / 23.1 ms = 0.09 / 0.36 / 0.49 of the fp64 matrix peak; where the rest goes
// (tools/ard_stamps.py, knock-out builds): DESIGN.md section 4a.
namespace fused {

// Timing-only variants (wrong results; tools/build_variant.sh ... "-DGPBO_DIAGNOSTICS -DGPBO_ARD_SKIP=bits"): what each phase
// costs the whole launch - 1: no elimination of the diagonal block, 2: no kernel entries, 4: no second product,
// 8: no first product.  Never in the shipped library.
#if defined(GPBO_DIAGNOSTICS) && defined(GPBO_ARD_SKIP)
constexpr int SKIP = GPBO_ARD_SKIP;
#else
constexpr int SKIP = 0;
#endif

#if defined(GPBO_DIAGNOSTICS) && defined(GPBO_ARD_STAMPS)
// in-kernel timeline of workgroup 0's first cell: (tag << 56 | s_memtime) words per wave, tools/ard_stamps.py
__device__ unsigned long long g_stamps[4][2048];
__device__ int g_nstamps[4];
#define ARD_STAMP(TAG)                                                                                                  \
    do {                                                                                                               \
        if (blockIdx.x == 0 && g == 0 && (threadIdx.x & 63) == 0) {                                                    \
            const int n_ = g_nstamps[w];                                                                               \
            if (n_ < 2048) { g_stamps[w][n_] = ((unsigned long long)(TAG) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffull); g_nstamps[w] = n_ + 1; } \
        }                                                                                                              \
    } while (0)
#else
#define ARD_STAMP(TAG) do { } while (0)
#endif

constexpr int OCC = 2;                  // workgroups (= waves per SIMD) per CU: LDS (the elimination's image) allows two
constexpr int TH = 256;                 // 4 waves, one per SIMD; several workgroups per CU fill each other's serial phases
constexpr int WAVES = TH / 64;
constexpr int LDM = gpbo_pd::LDM;       // 66
constexpr int SLOTS = 512;              // most workgroups of one launch = scratch slots (2 per CU on 256 CUs)
constexpr int DMAX = 16;                // widest feature bucket (GPBO_MAX_D)
constexpr int64_t WORK_CAP = 16ll << 30;

// what depends on the value is recomputed where it is used instead of being kept in registers across the whole kernel
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// The elimination of the diagonal block as a real call (M sits at the start of the dynamic LDS): inlined into the
// persistent loops, its lane masks and per-lane addresses were hoisted to the kernel's entry and held there; a call
// costs one save of the accumulators per panel.
__device__ __attribute__((noinline)) int potrf_panel_call() {
    extern __shared__ double smem_[];
    return gpbo_pd::potrf_diag64_lds(smem_, threadIdx.x);
}

template <int D>
struct Lds {
    double M[2 * 64 * LDM];   // [D_jj ; I] -> [L_jj ; L_jj^-T]
    double Xc[64 * D];        // coordinates of the panel's 64 columns
    double yc[64];            // y of the panel's columns
    double tab[GPBO_EXP_E];   // exp_neg's table
    double red[2 * WAVES];    // final reductions
    int bad;
};

// Padded inputs, once per call: Xp [(Nf + 16) x D] (features beyond d and rows beyond N are zero), yp [Nf],
// il2p [G x D] = 1 / l^2 (0 for the padded features: they add fma(0, 0, acc) = acc to a distance).
__global__ __launch_bounds__(256) void nlml_prep_kernel(const double *__restrict__ X, const double *__restrict__ y, int N,
                                                        int d, int D, int Nf, const double *__restrict__ ls_cells,
                                                        int64_t G, double *__restrict__ Xp, double *__restrict__ yp,
                                                        double *__restrict__ il2p) {
    const int64_t nx = (int64_t)(Nf + 16) * D, nl = G * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nx + Nf + nl; e += (int64_t)gridDim.x * 256) {
        if (e < nx) {
            const int64_t r = e / D;
            const int k = (int)(e - r * D);
            Xp[e] = (r < N && k < d) ? X[r * d + k] : 0.0;
        } else if (e < nx + Nf) {
            const int64_t r = e - nx;
            yp[r] = (r < N) ? y[r] : 0.0;
        } else {
            const int64_t q = e - nx - Nf, g = q / D;
            const int k = (int)(q - g * D);
            double v = 0.0;
            if (k < d) {
                const double l = ls_cells[g * d + k];
                v = 1.0 / (l * l);
            }
            il2p[q] = v;
        }
    }
}

// MODE 0: the reference's likelihood, float32, log(exp(logdet)) (point_selector.py:117-119: np.log(np.linalg.det(K)));
// MODE 1: fp64, log det straight from the factor (no underflow), NaN when a pivot fails.
template <int D, int MODE, int RT /* row tiles per block of a wave: 2 (4 measured and not instantiated, see the kernel) */>
__global__ __launch_bounds__(TH, OCC) void nlml_fused_kernel(const double *__restrict__ Xp, const double *__restrict__ yp, int N,
                                                         int Nf, const double *__restrict__ il2p, int G, double jitter,
                                                         void *__restrict__ out_, double *scratch) {
    extern __shared__ double smem_[];
    Lds<D> &S = *reinterpret_cast<Lds<D> *>(smem_);
    const int tid0 = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int KP = Nf >> 3;            // k-group pairs per row tile
    const int RTY = Nf >> 4;           // the row tile that carries y (row 0 of it)
    const int nbf = Nf >> 6;
    d2_t *Lf = reinterpret_cast<d2_t *>(scratch) + (int64_t)blockIdx.x * (int64_t)(RTY + 1) * KP * 64;

    if (tid0 < GPBO_EXP_E) S.tab[tid0] = kExp2Tab256[tid0 * (256 / GPBO_EXP_E)];

    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        gpbo_syncthreads();   // the previous cell's reductions have been read
        double il2[D];     // wave-uniform (scalar loads)
#pragma unroll
        for (int k = 0; k < D; ++k) il2[k] = il2p[(int64_t)g * D + k];
        if (tid0 == 0) S.bad = 0;
        double logdet_t = 0.0, quad_t = 0.0;

        for (int j = 0; j < nbf; ++j) {
            const int J0 = j << 6;
            // (opaque: lane masks and per-lane addresses are recomputed per panel, not kept in registers across the kernel)
            const int tid = opaque(tid0);
            const int lane = tid & 63, l15_ = lane & 15, l4_ = lane >> 4;
            // the panel's columns: coordinates, y; the identity under the diagonal block
            for (int e = tid; e < 64 * D; e += TH) S.Xc[e] = Xp[(int64_t)J0 * D + e];
            if (tid < 64) S.yc[tid] = yp[J0 + tid];
            ARD_STAMP(1);
            for (int e = tid; e < 64 * 64; e += TH) S.M[(64 + (e >> 6)) * LDM + (e & 63)] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0;
            gpbo_syncthreads();
            ARD_STAMP(2);

            // rows of the panel in blocks of RT row tiles (32 or 64 rows): the first 64 / (16 RT) blocks = the diagonal block,
            // block nblk = y's tile.  Wave w takes blocks w, w + 4, ...; the elimination of the diagonal block sits between the
            // two products of every wave's FIRST block (all four waves pass through every iteration, with or without a block).
            // (RT = 4 - a 64 x 64 wave tile, a third less operand traffic per flop, 128 accumulator registers, two register
            //  sets only - was built and measured: with one pair ahead on both sides it won from 1,300 observations on (N = 1536 /
            //  2048: 75.3 / 167.7 against 78.5 / 180.7 ms), against RT = 2 with its pairs two steps ahead it loses everywhere
            //  (74.9 / 163.4 against 69.8 / 161.1): only RT = 2 is instantiated.)
            constexpr int DB = 4 / RT;                    // blocks of the diagonal block
            const int nblk = (Nf - J0) / (16 * RT);
            const int jt0 = J0 >> 4;
            const int nkp = J0 >> 3;
            const int nit = (nblk + WAVES) / WAVES;     // ceil((nblk + 1) / WAVES)
            for (int it = 0; it < nit; ++it) {
                const int b = w + WAVES * it;
                const bool has = (b <= nblk);
                const bool yblk = (b == nblk);
                const int rt0 = jt0 + RT * (has ? b : 0);
                int rtt[RT];                                // the block's row tiles (y's block: its one tile RT times, stored once)
#pragma unroll
                for (int t = 0; t < RT; ++t) rtt[t] = (yblk || !has) ? rt0 : rt0 + t;
                d4_t acc[4][RT];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int t = 0; t < RT; ++t) acc[ct][t] = d4_t{0.0, 0.0, 0.0, 0.0};
                if (nkp > 0 && !(SKIP & 8)) {
                    // ---- C' = K^T - L_j L_b^T: operands straight from memory, the next k-group pair in flight ----
                    const d2_t *pa = Lf + ((int64_t)jt0 * KP) * 64 + lane;    // + ct * KP * 64
                    const d2_t *pb[RT];
#pragma unroll
                    for (int t = 0; t < RT; ++t) pb[t] = Lf + ((int64_t)rtt[t] * KP) * 64 + lane;
                    const int64_t sa = (int64_t)KP * 64;
                    // two register sets in turn (nkp is a multiple of 8): the loads of one pair are in flight while the
                    // products of the other issue; the scheduling barriers keep that order
                    struct Frag { d2_t f[4], g[RT]; } A, B;
                    auto load = [&](Frag &F, int k) {
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) F.f[ct] = pa[ct * sa + k * 64];
#pragma unroll
                        for (int t = 0; t < RT; ++t)
                            if (t == 0 || !yblk) F.g[t] = pb[t][k * 64];
                    };
                    auto mult = [&](const Frag &F) {
#pragma unroll
                        for (int t = 0; t < RT; ++t)
                            if (t == 0 || !yblk) {
#pragma unroll
                                for (int ct = 0; ct < 4; ++ct) acc[ct][t] = mfma_f64_16x16x4(F.f[ct].x, F.g[t].x, acc[ct][t]);
                            }
#pragma unroll
                        for (int t = 0; t < RT; ++t)
                            if (t == 0 || !yblk) {
#pragma unroll
                                for (int ct = 0; ct < 4; ++ct) acc[ct][t] = mfma_f64_16x16x4(F.f[ct].y, F.g[t].y, acc[ct][t]);
                            }
                    };
                    // (Every wave fetches the panel rows - pa, the operand the four share - for itself: 104 GB per 2,500 cells
                    //  at N = 1024, d = 8 = 1.5x what the 64-column algorithm itself must move.  Sharing them was built (with ONE pair ahead)
                    //  three ways and measured slower each time, same box: slabs staged through LDS with one barrier per 64
                    //  products 23.5 against 22.7 ms, three 8-KB stages three slabs ahead with one barrier per 32 products
                    //  25.6 against 22.6, a bare barrier every 16 / 32 / 64 / 128 products so that the second to fourth
                    //  reader hit L2 23.3 / 23.1 / 23.0 / 22.9 against 22.7: a wave's pace depends on what the OTHER
                    //  workgroup's wave on its SIMD is doing, and every meeting point makes the four wait for the slowest.)
                    if (has) {
                        // four register sets in rotation, every pair TWO steps in front of its products (nkp is a multiple of
                        // 8): with one pair ahead the loop waited for memory, not for the matrix pipes, whenever the other
                        // workgroup's wave left this SIMD's pipe to it - 2,500 cells at N = 512 / 1024, same box, d = 2:
                        // 4.00 / 24.7 ms one ahead, 3.63 / 22.6 two, 3.65 / 22.6 three
                        Frag C, E;
                        auto ld = [&](Frag &F, int k) { load(F, k < nkp ? k : nkp - 1); };   // (the tail re-loads the last pair)
                        load(A, 0);
                        load(B, 1);
                        for (int kp = 0; kp < nkp; kp += 4) {
                            ld(C, kp + 2);
                            __builtin_amdgcn_sched_barrier(0);
                            mult(A);
                            __builtin_amdgcn_sched_barrier(0);
                            ld(E, kp + 3);
                            __builtin_amdgcn_sched_barrier(0);
                            mult(B);
                            __builtin_amdgcn_sched_barrier(0);
                            ld(A, kp + 4);
                            __builtin_amdgcn_sched_barrier(0);
                            mult(C);
                            __builtin_amdgcn_sched_barrier(0);
                            ld(B, kp + 5);
                            __builtin_amdgcn_sched_barrier(0);
                            mult(E);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if (has) {
                    ARD_STAMP(3);
                    // K's entries: register r of tile (ct, t) of lane l is column J0 + 16 ct + l4 + 4 r, row 16 rt_t + l15
                    const int l15 = opaque(l15_), l4 = opaque(l4_);
                    const bool interior = b >= DB && !yblk && (rtt[RT - 1] << 4) + 15 < N && J0 + 63 < N;
                    if (SKIP & 2) {
                    } else if (yblk) {       // y's tile: row 0 carries y, no kernel entry at all
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const double v = (l15 == 0) ? S.yc[16 * ct + l4 + 4 * r] : 0.0;
                                acc[ct][0][r] = v - acc[ct][0][r];
                            }
                    } else if (interior) {   // every row and column is an observation, no diagonal entry
#pragma unroll
                        for (int t = 0; t < RT; ++t) {
                            const double *xp = Xp + (int64_t)((rtt[t] << 4) + l15) * D;
                            double xr[D];
#pragma unroll
                            for (int k = 0; k < D; ++k) xr[k] = xp[k];
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const double *xc = S.Xc + (16 * ct + l4 + 4 * r) * D;
                                    double a = 0.0;
#pragma unroll
                                    for (int k = 0; k < D; ++k) {
                                        const double diff = xc[k] - xr[k];
                                        a = fma(diff * diff, il2[k], a);
                                    }
                                    acc[ct][t][r] = exp_neg(0.5 * a, S.tab) - acc[ct][t][r];
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    } else {          // the diagonal block, rows / columns of the padding
#pragma unroll
                        for (int t = 0; t < RT; ++t) {
                            const int row = (rtt[t] << 4) + l15;
                            const double *xp = Xp + (int64_t)row * D;
                            double xr[D];
#pragma unroll
                            for (int k = 0; k < D; ++k) xr[k] = xp[k];
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int jc = 16 * ct + l4 + 4 * r, col = J0 + jc;
                                    double a = 0.0;
#pragma unroll
                                    for (int k = 0; k < D; ++k) {
                                        const double diff = S.Xc[jc * D + k] - xr[k];
                                        a = fma(diff * diff, il2[k], a);
                                    }
                                    double v = exp_neg(0.5 * a, S.tab);
                                    if (row == col) v += jitter;
                                    if (row >= N || col >= N) v = (row == col) ? 1.0 : 0.0;
                                    acc[ct][t][r] = v - acc[ct][t][r];
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                }
                ARD_STAMP(4);
                if (it == 0) {
                    if (has && b < DB) {   // the diagonal block's rows: lower triangle into LDS
                        const int l15 = opaque(l15_), l4 = opaque(l4_);
#pragma unroll
                        for (int t = 0; t < RT; ++t)
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int lrow = 16 * RT * b + 16 * t + l15, jc = 16 * ct + l4 + 4 * r;
                                    S.M[lrow * LDM + jc] = (jc <= lrow) ? acc[ct][t][r] : 0.0;
                                }
                    }
                    gpbo_syncthreads();
                    ARD_STAMP(5);
                    const int fbad = (SKIP & 1) ? 0 : potrf_panel_call();
                    ARD_STAMP(6);
                    if (tid == 0 && fbad) S.bad = 1;
                    if (tid < 64) logdet_t += log(S.M[tid * LDM + tid]);
                }
                ARD_STAMP(7);
                if (has && b >= DB && !(SKIP & 4)) {
                    // ---- L^T = W C': stored in fragment order; y's row adds its squares to |z|^2 ----
                    const int l15 = opaque(l15_), l4 = opaque(l4_);
#pragma unroll
                    for (int cq = 0; cq < 4; ++cq) {        // output column tile: panel columns 16 cq ..
                        d4_t o[RT];
#pragma unroll
                        for (int t = 0; t < RT; ++t) o[t] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int ct = 0; ct <= cq; ++ct)    // W is lower triangular
#pragma unroll
                            for (int kg = 0; kg < 4; ++kg) {
                                const double a = S.M[(64 + 16 * ct + 4 * kg + l4) * LDM + 16 * cq + l15];   // W[16cq + l15][16ct + 4kg + l4]
#pragma unroll
                                for (int t = 0; t < RT; ++t)
                                    if (t == 0 || !yblk) o[t] = mfma_f64_16x16x4(a, acc[ct][t][kg], o[t]);
                            }
                        const int kp0 = nkp + 2 * cq;
#pragma unroll
                        for (int t = 0; t < RT; ++t)
                            if (t == 0 || !yblk) {
                                d2_t *q = Lf + ((int64_t)rtt[t] * KP + kp0) * 64 + lane;
                                q[0] = d2_t{o[t][0], o[t][1]};
                                q[64] = d2_t{o[t][2], o[t][3]};
                            }
                        if (yblk && l15 == 0) {
                            quad_t = fma(o[0][0], o[0][0], quad_t); quad_t = fma(o[0][1], o[0][1], quad_t);
                            quad_t = fma(o[0][2], o[0][2], quad_t); quad_t = fma(o[0][3], o[0][3], quad_t);
                        }
                    }
                }
            }
            ARD_STAMP(8);
            gpbo_syncthreads();   // the panel's fragments are visible to every wave; M and Xc are free
            ARD_STAMP(9);
        }

        // log det K = 2 sum log L_ii (threads 0..63 = wave 0), |z|^2 from whichever waves owned y's tile
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            logdet_t += __shfl_xor(logdet_t, off);
            quad_t += __shfl_xor(quad_t, off);
        }
        if ((tid0 & 63) == 0) { S.red[w] = logdet_t; S.red[WAVES + w] = quad_t; }
        gpbo_syncthreads();
        if (tid0 == 0) {
            double quad = 0.0;
            for (int k = 0; k < WAVES; ++k) quad += S.red[WAVES + k];
            const double logdet = 2.0 * S.red[0];
            double nlml;
            if (MODE == 0) {
                const double logdet_ref = log(exp(logdet));  // the reference takes log of a det that under/overflows
                nlml = 0.5 * (quad + logdet_ref + (double)N * 1.8378770664093453);  // log(2 pi)
            } else {
                nlml = 0.5 * (quad + logdet + (double)N * 1.8378770664093453);
            }
            if (S.bad) nlml = __builtin_nan("");   // not positive definite: the reference's log(det < 0) is NaN
            if (MODE == 0) reinterpret_cast<float *>(out_)[g] = (float)nlml;
            else reinterpret_cast<double *>(out_)[g] = nlml;
        }
    }
}

inline int64_t slot_bytes(int64_t Nf) { return (Nf + 16) * Nf * 8; }
inline int64_t slots_for(int64_t Nf, int64_t G) {
    int64_t s = WORK_CAP / slot_bytes(Nf);
    if (s > SLOTS) s = SLOTS;
    if (s > G) s = G;
    return s < 1 ? 1 : s;
}
// padded inputs in front of the slots: Xp, yp, il2p (sized for the widest bucket), rounded up to 256 bytes
inline int64_t head_bytes(int64_t Nf, int64_t G) {
    const int64_t b = ((Nf + 16) * DMAX + Nf + G * DMAX) * 8;
    return (b + 255) / 256 * 256;
}

template <int D, int MODE, int RT>
int launch_rt(const double *Xp, const double *yp, int64_t N, int64_t Nf, const double *il2p, int64_t G, double jitter, void *out,
              double *scratch, hipStream_t st) {
    const void *fn = reinterpret_cast<const void *>(nlml_fused_kernel<D, MODE, RT>);
    const size_t lds = sizeof(Lds<D>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GPBO_ERR_LAUNCH;
    // as many persistent workgroups as are resident at once (registers and LDS decide), never more than there are slots
    static std::atomic<int> resident_{0};   // per template instance; the device's answer does not change
    int resident = resident_.load(std::memory_order_relaxed);
    if (resident == 0) {
        int per_cu = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, TH, lds) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
        resident = per_cu * cus;
        resident_.store(resident, std::memory_order_relaxed);
    }
    int64_t grid = slots_for(Nf, G);
    if (grid > resident) grid = resident;
    hipLaunchKernelGGL((nlml_fused_kernel<D, MODE, RT>), dim3((unsigned)grid), dim3(TH), lds, st, Xp, yp, (int)N, (int)Nf,
                       il2p, (int)G, jitter, out, scratch);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

template <int D, int MODE>
int launch(const double *Xp, const double *yp, int64_t N, int64_t Nf, const double *il2p, int64_t G, double jitter, void *out,
           double *scratch, hipStream_t st) {
    return launch_rt<D, MODE, 2>(Xp, yp, N, Nf, il2p, G, jitter, out, scratch, st);
}

template <int MODE>
int run(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells, int64_t G, double jitter, void *out,
        void *work, int64_t work_bytes, void *stream) {
    if (!X || !y || !ls_cells || !out || !work || N < 1 || N > (1 << 15) || d < 1 || d > GPBO_MAX_D || G < 1 || G > (1 << 30))
        return GPBO_ERR_ARG;
    const int64_t Nf = (N + GPBO_NB - 1) / GPBO_NB * GPBO_NB;
    if (work_bytes < head_bytes(Nf, G) + slots_for(Nf, G) * slot_bytes(Nf) || ((uintptr_t)work & 255)) return GPBO_ERR_WORKSPACE;
    hipStream_t st = gpbo_stream(stream);
    const int D = d <= 2 ? 2 : d <= 4 ? 4 : d <= 8 ? 8 : 16;
    double *Xp = reinterpret_cast<double *>(work);
    double *yp = Xp + (Nf + 16) * D;
    double *il2p = yp + Nf;
    double *scratch = reinterpret_cast<double *>(reinterpret_cast<char *>(work) + head_bytes(Nf, G));
    const int64_t total = (Nf + 16) * D + Nf + G * D;
    const unsigned pg = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(nlml_prep_kernel, dim3(pg), dim3(256), 0, st, X, y, (int)N, (int)d, D, (int)Nf, ls_cells, G, Xp, yp, il2p);
    GPBO_CHECK_LAUNCH();
    if (D == 2) return launch<2, MODE>(Xp, yp, N, Nf, il2p, G, jitter, out, scratch, st);
    if (D == 4) return launch<4, MODE>(Xp, yp, N, Nf, il2p, G, jitter, out, scratch, st);
    if (D == 8) return launch<8, MODE>(Xp, yp, N, Nf, il2p, G, jitter, out, scratch, st);
    return launch<16, MODE>(Xp, yp, N, Nf, il2p, G, jitter, out, scratch, st);
}

}  // namespace fused

}  // namespace

extern "C" int64_t gpbo_nlml_grid_batched_workspace_bytes(int64_t N, int64_t G) {
    if (N < 1 || G < 1 || N > (1 << 15)) return GPBO_ERR_ARG;
    const int64_t Nf = (N + GPBO_NB - 1) / GPBO_NB * GPBO_NB;
    return fused::head_bytes(Nf, G) + fused::slots_for(Nf, G) * fused::slot_bytes(Nf);
}

extern "C" int gpbo_nlml_grid_batched_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_cells,
                                          int64_t G, double jitter, float *out, void *work, int64_t work_bytes,
                                          void *stream) {
    return fused::run<0>(X, y, N, d, ls_cells, G, jitter, out, work, work_bytes, stream);
}

extern "C" int gpbo_nlml_grid_batched_logdet_f64(const double *X, const double *y, int64_t N, int32_t d,
                                                 const double *ls_cells, int64_t G, double jitter, double *out, void *work,
                                                 int64_t work_bytes, void *stream) {
    return fused::run<1>(X, y, N, d, ls_cells, G, jitter, out, work, work_bytes, stream);
}

#if defined(GPBO_DIAGNOSTICS) && defined(GPBO_ARD_STAMPS)
extern "C" int gpbo_diag_ard_stamps(unsigned long long *out, int *counts) {   // host buffers [4][2048], [4]
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fused::g_stamps), sizeof(unsigned long long) * 4 * 2048) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(counts, HIP_SYMBOL(fused::g_nstamps), sizeof(int) * 4) != hipSuccess) return -1;
    int z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(fused::g_nstamps), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

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
