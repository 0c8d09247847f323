// K(X*,X) entries and the mean with the pair distances on the fp64 matrix cores - for the PREFIX-BOUND route only
// (sigma_acq.hip: gpbo_posterior_prefix_f64), where the first pass needs a BOUND, not the reference's bits.
//
//     t_ci = |a_c - b_i|^2 = |a_c|^2 + |b_i|^2 - 2 a_c . b_i,        k_ci = exp(-t_ci)           (point_selector.py:180-189)
// with a = (x* - m) / (ls sqrt 2), b = (x - m) / (ls sqrt 2), m = the mean of the scaled observations (centring keeps the
// norms, hence the cancellation, small).  The expanded form is an inner product of length d on top of the two norms,
//     t_ci = (|a_c|^2 + |b_i|^2) + b_i . (-2 a_c),
// i.e. ceil(d / 4) v_mfma_f64_16x16x4_f64 per 16 x 16 block of pairs (the norms are the accumulator's initial value: one
// VALU addition per pair) instead of 2 d fp64 VALU instructions per pair;
// what stays on the VALU is exp(-t) and the mean's multiply-add (25 of the 36 instructions of kstar_mu_kernel; 19 with the
// shorter exp a bound can afford: 256-entry table, degree 4, one-word argument reduction, clamp instead of compare-select).
// Measured (MI355X, N = 4096, rocprofv3 kernel trace), ms per 2^17 x 4096 entries: difference-form kernel without its
// stores 0.61; this kernel with the full-length exp and the norms inside the inner product (3 MFMAs) 0.57-0.61 - no gain;
// with the shorter exp 0.53; with the norms as the accumulator's initial value (2 MFMAs) 0.48-0.51.  The accounting that
// fits: v_mfma_f64_16x16x4_f64 occupies a SIMD for 64 cycles (1,024 multiply-adds at the 16 per cycle behind 78.6 TFLOP/s)
// and fp64 MFMA does not execute beside fp64 VALU on gfx950 (K(X*,X) overlapped with the variance kernel
// gained nothing), so per pair and lane the MFMAs cost 32 cycles (48 with three) where the sixteen VALU instructions they
// replace cost 64, and the rest of the pair 19 x 4 = 76 (25 x 4 = 100 with the full exp): 108 cycles against 164 for the
// difference form - measured 130 against 150.  Issue order does not matter (operands from LDS instead of L2, branch-free
// stores, three-address FMAs in asm, the four exp chains of a tile in lock step: no change).  The route also gained from this
// kernel's coarser partials of the mean: 16 instead of 64 per candidate for the first-pass variance launches to add up
// (0.21 -> 0.18 ms each).
//
// Accuracy - why this is NOT the fp64 path's kernel.  The difference form's error in t is relative (t 1e-16); the expanded
// form's is ABSOLUTE: every one of the d + 2 fused steps rounds at the size of the partial sum, <= 2 (|a|^2 + |b|^2), so
//     |dt_ci| <= gamma (|a_c|^2 + |b_i|^2),   gamma = 2 (2 (d + 2) + 8) 2^-53   (inputs' own roundings included, x2 to spare)
// and since |dk| <= k |dt| and k <= 1, the mean  mu_c = sum_i k_ci alpha_i  is off by at most
//     slack_c = gamma (|a_c|^2 S0 + S1) + (eps_exp + eps_acc) S0,      S0 = sum_i |alpha_i|,   S1 = sum_i |alpha_i| |b_i|^2
// (eps_exp: the error of this file's shorter exp, see exp_neg_m; eps_acc: the rounding of the two kernels' summations)
// (N = 4096 benchmark problem: S0 = 7.8e4, slack ~ 1e-9; the 5e-15 entry bound of tests/test_kstar_mu cannot be met).
// The prefix-bound route only needs mu from BELOW (both acquisitions decrease with the mean): this kernel reports
// mu_c - slack_c, so the upper bound it feeds stays an upper bound; the entries it stores (first `store_rows` observations)
// feed |v[:J]|^2, whose error (<= 3e-10) is inside the variance pad GPBO_BOUND_VAR_PAD.  The candidates that survive
// are re-scored by the difference-form kernel (kernel_build.hip) - the answer never contains a number from this file.
//
// Layout.  Workgroup = 4 waves, wave = 64 candidates (4 MFMA column tiles) x one slice of 64 observations (4 row tiles):
// output register r of lane (l15, l4) is the pair (observation 16 ot + l4 + 4 r, candidate 16 tc + l15), so a lane keeps
// the mean of ITS candidate over its rows and the four lanes of a candidate are added once at the end; K*^T rows are
// stored as 128-byte segments (16 candidates).  mu_part is [Np / slice][ldk] with slice = gpbo_kstar_mfma_slice(Np)
// observations per workgroup (kstar_mu_kernel's layout with a coarser slice).
#include "gpbo_internal.h"
#include "exp_neg.h"

#include <type_traits>

namespace {

constexpr int KP_MAX = 20;                 // d + 2 <= 18, padded to a multiple of 4
constexpr int KS = GPBO_KS_SLICE;          // observations per slice (64)
constexpr int OB_MAX = 256;                // observations per workgroup (LDS: 256 x 21 doubles)
constexpr double PAD_ROW_NORM = 1e300;     // |b|^2 of padding rows i >= N: t = 1e300, k = 0 exactly

struct ObsPrep {
    double m[GPBO_MAX_D];   // centre (scaled coordinates)
    double S0, S1;
};

struct IscArgs {
    double isc[GPBO_MAX_D];  // 1 / (ls_k sqrt 2)
};

// 2^(j/256), j = 0 .. 255: kExp2Tab256 of exp_neg.h (one table for every kernel of the library)

// exp(-t) for 0 <= t <= 708, for THIS file's purpose (a bound with its error carried along - not the fp64 path's exp_neg):
// 2^(n/256) table x degree-4 polynomial; |r| <= ln2/512 leaves r^5/120 < 4e-17 relative, and reducing with the leading
// double of ln2/256 only costs |n| 3e-19 in r, i.e. at most t e^-t 369 x 3e-19 <= 4e-17 ABSOLUTE in the result: both inside
// the EXP_ABS_ERR / EXP_REL_ERR that the caller adds to its slack.  19 instead of 25 VALU instructions per pair with the
// clamp below instead of the compare-and-select pair.
constexpr double EXP_REL_ERR = 4.0 * 1.1102230246251565e-16, EXP_ABS_ERR = 1e-16;
// Round 4: the SUMMATION of the mean is rounded too - here (64 products per lane, two shuffles, Np / 256 partials) and in the
// plain pass this mean must stay below (kstar_mu_kernel: 64 products per slice, Np / 64 partials).  |fl(sum) - sum| <=
// n 2^-53 sum |terms| <= n 2^-53 S0 for each chain of n additions: eps_acc = (2 x 64 + 4 + Np / 64 + Np / 256) x 2^-53 x 2
// (gpbo_kstar_mu_mfma).  Without it the report "from below" rested on the distance term alone, which vanishes for a
// candidate at the centroid of observations that all lie within a length scale of it.
__device__ __forceinline__ double exp_neg_m(double t, const double *tab) {
    const double z = fma(-t, 369.3299304675746, 6755399441055744.0);
    const int ni = __double2loint(z);
    const double fn = z - 6755399441055744.0;
    const double r = fma(fn, -0.0027076061740622863, -t);
    const double T = tab[ni & 255];
    double q = fma(r, 1.0 / 24.0, 1.0 / 6.0);
    q = fma(r, q, 0.5);
    q = fma(r, q, 1.0);
    const double v = fma(T * r, q, T);             // in [1, 2): scale by 2^(n >> 8) through the exponent field
    return __hiloint2double(__double2hiint(v) + ((ni >> 8) << 20), __double2loint(v));
}

// Round 5: the entries that are NOT stored (15/16 of them: they only feed the mean, which is reported from below anyway) take a
// shorter exponential still: a 1,024-entry table 2^(j/1024) in LDS and ONE term of the series, exp(r) ~ 1 + r with
// |r| <= ln2 / 2048: relative error r^2 / 2 <= 5.8e-8 (+ 1e-13 from the one-word reduction), carried in the slack as
// EXP_FAST_REL_ERR S0 (4.7e-3 on the N = 4096 benchmark problem, S0 = 7.8e4: the first level keeps 14,659 instead of 14,206
// of 2^21 candidates against the best exact value - tools/bound_fp32_probe.py - where an fp32 exponential, 1.5e-7 S0, keeps
// 14,749 at more instructions: v_cvt_f32_f64 / v_exp_f32 / v_cvt_f64_f32 against the three fused operations saved here).
// 10 instead of 19 VALU instructions per pair; the stored entries keep exp_neg_m (they feed |v[:J]|^2, whose pad is 1e-8).
constexpr double EXP_FAST_REL_ERR = 6.0e-8;
__device__ __forceinline__ double exp_neg_fast(double t, const double *tab1024) {
    double u;
    asm("v_min_f64 %0, %1, %2" : "=v"(u) : "v"(t), "v"(708.0));   // (padding rows: t = 1e300; NaN -> 708; t < 0 by a rounding: fine)
    const double z = fma(-u, 1477.3197218702985, 6755399441055744.0);    // 1024 / ln 2
    const int ni = __double2loint(z);
    const double fn = z - 6755399441055744.0;
    const double r = fma(fn, -6.7690154351557157e-4, -u);                 // ln 2 / 1024
    const double T = tab1024[ni & 1023];
    const double v = fma(T, r, T);                 // in [1, 2): scale by 2^(n >> 10) through the exponent field
    return __hiloint2double(__double2hiint(v) + ((ni >> 10) << 20), __double2loint(v));
}

// clamp to [0, 708] without the canonicalising v_max the compiler puts in front of fmax / fmin on an MFMA result (a rounding
// can leave a tiny negative distance; beyond 708 the result is below 1e-307 either way; NaN -> 0: the caller poisons the mean)
__device__ __forceinline__ double clamp_t(double t) {
    double u;
    asm("v_max_f64 %0, %1, 0" : "=v"(u) : "v"(t));
    asm("v_min_f64 %0, %1, %2" : "=v"(u) : "v"(u), "v"(708.0));
    return u;
}

// One workgroup: centre, rows (b_1 .. b_d, 1, |b|^2, 0 ..) of the observations, S0, S1.  N x d is small (4096 x 8).
__global__ __launch_bounds__(1024) void obs_prep_kernel(const double *__restrict__ X, int N, int Np, int d, IscArgs ls,
                                                        const double *__restrict__ alpha, int KP, double *__restrict__ Bp,
                                                        ObsPrep *__restrict__ prep) {
    __shared__ double red[1024];
    __shared__ double m_s[GPBO_MAX_D];
    const int t = threadIdx.x;
    for (int k = 0; k < d; ++k) {
        double s = 0.0;
        for (int i = t; i < N; i += 1024) s += X[(int64_t)i * d + k] * ls.isc[k];
        red[t] = s;
        __syncthreads();
        for (int h = 512; h > 0; h >>= 1) {
            if (t < h) red[t] += red[t + h];
            __syncthreads();
        }
        if (t == 0) m_s[k] = red[0] / (double)N;
        __syncthreads();
    }
    double s0 = 0.0, s1 = 0.0;
    for (int i = t; i < Np; i += 1024) {
        double *row = Bp + (int64_t)i * KP;
        if (i < N) {
            double nb = 0.0;
            for (int k = 0; k < d; ++k) {
                const double b = fma(X[(int64_t)i * d + k], ls.isc[k], -m_s[k]);
                row[k] = b;
                nb = fma(b, b, nb);
            }
            row[d] = 1.0;
            row[d + 1] = nb;
            for (int k = d + 2; k < KP; ++k) row[k] = 0.0;
            const double aa = fabs(alpha[i]);
            s0 += aa;
            s1 = fma(aa, nb, s1);
        } else {
            for (int k = 0; k < KP; ++k) row[k] = 0.0;
            row[d + 1] = PAD_ROW_NORM;
        }
    }
    red[t] = s0;
    __syncthreads();
    for (int h = 512; h > 0; h >>= 1) {
        if (t < h) red[t] += red[t + h];
        __syncthreads();
    }
    const double S0 = red[0];
    __syncthreads();
    red[t] = s1;
    __syncthreads();
    for (int h = 512; h > 0; h >>= 1) {
        if (t < h) red[t] += red[t + h];
        __syncthreads();
    }
    if (t == 0) {
        for (int k = 0; k < GPBO_MAX_D; ++k) prep->m[k] = k < d ? m_s[k] : 0.0;
        prep->S0 = S0;
        prep->S1 = red[0];
    }
}

// (MFMA results straight into VGPRs - `-mllvm -amdgpu-mfma-vgpr-form=1`, no v_accvgpr_read per value - was measured too:
//  13.2-13.4 against 13.3-13.5 ms per 2^21 candidates for the whole route, at 128-130 registers instead of 94: not used.)
template <int KQ /* ceil(d / 4): MFMAs per 16 x 16 pairs */>
__global__ __launch_bounds__(256) void kstar_mu_mfma_kernel(const double *__restrict__ Xs, int64_t Mc, int d, IscArgs ls,
                                                            const double *__restrict__ Bp, const ObsPrep *__restrict__ prep,
                                                            const double *__restrict__ alpha, int N, double gamma,
                                                            double eps_acc /* rounding of the means' summations */,
                                                            double *__restrict__ KsT, int64_t ldk,
                                                            double *__restrict__ mu_part, int store_rows,
                                                            int OB /* observations per workgroup = per mean partial */,
                                                            int KP /* row length of Bp: d + 2 padded to a multiple of 4 */) {
    constexpr int NC = 4;   // candidate tiles per wave
    const int LDB = KP + 1;   // LDS row stride of the observation rows (odd: the 16 rows of a tile land on different banks)
    __shared__ double tab[256];
    __shared__ double tabf[1024];                  // 2^(j/1024) = 2^((j >> 2)/256) x 2^((j & 3)/1024): exp_neg_fast
    __shared__ double Bs[OB_MAX * (4 * KQ + 5)];   // KP <= 4 KQ + 4
    __shared__ double As[OB_MAX];   // alpha of the workgroup's observations (0 beyond N: those rows give k = 0 anyway)
    tab[threadIdx.x] = kExp2Tab256[threadIdx.x];   // (256 threads)
    {
        const double t0 = kExp2Tab256[threadIdx.x];
        tabf[4 * threadIdx.x + 0] = t0;
        tabf[4 * threadIdx.x + 1] = t0 * 1.0006771306930664;    // 2^(1/1024)
        tabf[4 * threadIdx.x + 2] = t0 * 1.0013547198921082;    // 2^(2/1024)
        tabf[4 * threadIdx.x + 3] = t0 * 1.002032767907594;    // 2^(3/1024)
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t cbase = (int64_t)blockIdx.x * 256 + wid * 64;
    const int n0 = blockIdx.y * OB;
    // the workgroup's observation rows, once: the MFMA operands then come from LDS (a global load per operand and tile left
    // every wave waiting on L2 at the top of every tile)
    for (int e = threadIdx.x; e < OB * KP; e += 256) {
        const int row = e / KP, k = e - row * KP;
        Bs[row * LDB + k] = Bp[(int64_t)(n0 + row) * KP + k];
    }
    for (int e = threadIdx.x; e < OB; e += 256) As[e] = (n0 + e < N) ? alpha[n0 + e] : 0.0;
    // candidate operands: element e = 4 q + l4 of (-2 a_1 .. -2 a_d, 0 ..) for candidate cbase + 16 tc + l15.  The two norms
    // do not ride in the inner product (that would be a third MFMA at d = 8 - 64 cycles - for two additions): |a|^2 + |b|^2
    // is the accumulator's initial value.  (The observation rows keep their (1, |b|^2) tail: it meets zeros here.)
    double bop[NC][KQ], slack[NC], nac[NC];
    bool isnan_c[NC];
#pragma unroll
    for (int tc = 0; tc < NC; ++tc) {
        const int64_t c = cbase + 16 * tc + l15;
        double na = 0.0;
        double sel[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) sel[q] = 0.0;
        bool bad = false;
        for (int k = 0; k < d; ++k) {
            const double x = (c < Mc) ? Xs[c * d + k] : 0.0;
            bad = bad || (x != x);
            const double a = fma(x, ls.isc[k], -prep->m[k]);
            na = fma(a, a, na);
#pragma unroll
            for (int q = 0; q < KQ; ++q)
                if (4 * q + l4 == k) sel[q] = -2.0 * a;
        }
#pragma unroll
        for (int q = 0; q < KQ; ++q) bop[tc][q] = sel[q];
        // an infinite (or overflowing) coordinate makes the expanded form inf - inf; the difference form handles it (k = 0):
        // such a candidate is reported like a NaN one here - its bound is NaN, so it always survives to the fp64 kernels
        bad = bad || !(na < __builtin_inf());
        nac[tc] = na;
        slack[tc] = gamma * fma(na, prep->S0, prep->S1) + (EXP_FAST_REL_ERR + EXP_REL_ERR + EXP_ABS_ERR + eps_acc) * prep->S0;
        isnan_c[tc] = bad;
    }
    __syncthreads();

    double mu[NC];
#pragma unroll
    for (int tc = 0; tc < NC; ++tc) mu[tc] = 0.0;

    // one tile of 16 observations against the wave's 64 candidates; STORE: this tile's K*^T rows are kept
    auto tile = [&](int ot, auto store_c) {
        constexpr bool STORE = decltype(store_c)::value;
        const int r0 = n0 + 16 * ot;
        // observation operand: element 4 q + l4 of row r0 + l15
        double aop[KQ];
        const double *brow = Bs + (16 * ot + l15) * LDB + l4;
#pragma unroll
        for (int q = 0; q < KQ; ++q) aop[q] = brow[4 * q];
        double al[4], nb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            al[r] = As[16 * ot + l4 + 4 * r];
            nb[r] = Bs[(16 * ot + l4 + 4 * r) * LDB + d + 1];
        }
#pragma unroll
        for (int tc = 0; tc < NC; ++tc) {
            d4_t acc = {nac[tc] + nb[0], nac[tc] + nb[1], nac[tc] + nb[2], nac[tc] + nb[3]};
#pragma unroll
            for (int q = 0; q < KQ; ++q) acc = mfma_f64_16x16x4(aop[q], bop[tc][q], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double k = STORE ? exp_neg_m(clamp_t(acc[r]), tab) : exp_neg_fast(acc[r], tabf);
                mu[tc] = fma(k, al[r], mu[tc]);
                if (STORE) KsT[(int64_t)(r0 + l4 + 4 * r) * ldk + cbase + 16 * tc + l15] = k;
            }
        }
    };
    int ot_store = (store_rows - n0) / 16;   // store_rows and n0 are multiples of 64
    if (ot_store < 0) ot_store = 0;
    if (ot_store > OB / 16) ot_store = OB / 16;
    int ot = 0;
#pragma unroll 1
    for (; ot < ot_store; ++ot) tile(ot, std::true_type{});
#pragma unroll 1
    for (; ot < OB / 16; ++ot) tile(ot, std::false_type{});
#pragma unroll
    for (int tc = 0; tc < NC; ++tc) {
        double v = mu[tc];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (l4 == 0) {
            if (blockIdx.y == 0) v -= slack[tc];   // the mean from BELOW: see the header
            mu_part[(int64_t)blockIdx.y * ldk + cbase + 16 * tc + l15] = isnan_c[tc] ? __builtin_nan("") : v;
        }
    }
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

}  // namespace

// observations per workgroup = per partial of the mean (mu_part has Np / slice rows): the candidate operands of a wave
// (its share of the prologue) are then used for 256 observations where Np allows, not for 64
int gpbo_kstar_mfma_slice(int64_t Np) { return (Np % 256 == 0) ? 256 : 128; }

int64_t gpbo_kstar_mfma_prep_bytes(int64_t Np) {
    return align_up((int64_t)sizeof(double) * Np * KP_MAX, 256) + align_up((int64_t)sizeof(ObsPrep), 256);
}

// once per call: centred observation rows, S0, S1 (alpha: the factorisation's, padded with zeros to Np)
int gpbo_kstar_mfma_prep(const double *X, int64_t N, int64_t Np, int32_t d, const double *ls_host, const double *alpha,
                         void *prep_buf, void *stream) {
    if (!X || !ls_host || !alpha || !prep_buf || N < 1 || Np < N || d < 1 || d > GPBO_MAX_D) return GPBO_ERR_ARG;
    IscArgs ls;
    for (int k = 0; k < GPBO_MAX_D; ++k) ls.isc[k] = 0.0;
    for (int k = 0; k < d; ++k) {
        if (!(ls_host[k] > 0.0)) return GPBO_ERR_ARG;
        ls.isc[k] = 1.0 / (ls_host[k] * 1.4142135623730950488);
    }
    const int KP = (d + 2 + 3) / 4 * 4;
    double *Bp = reinterpret_cast<double *>(prep_buf);
    ObsPrep *prep = reinterpret_cast<ObsPrep *>(reinterpret_cast<char *>(prep_buf) +
                                                align_up((int64_t)sizeof(double) * Np * KP_MAX, 256));
    hipLaunchKernelGGL(obs_prep_kernel, dim3(1), dim3(1024), 0, gpbo_stream(stream), X, (int)N, (int)Np, (int)d, ls, alpha, KP,
                       Bp, prep);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

// K*^T rows n < store_rows (a multiple of 64) and the mean partials of all observations for Mc candidates: the interface of
// gpbo_kstar_mu_rows, the mean reported from BELOW by the error bound of the expanded distance (see the header).
int gpbo_kstar_mu_mfma(const double *Xs, int64_t Mc, int64_t N, int64_t Np, int32_t d, const double *ls_host,
                       const double *alpha, const void *prep_buf, double *KsT, int64_t ldk, double *mu_part,
                       int64_t store_rows, void *stream) {
    if (!Xs || !ls_host || !alpha || !prep_buf || !KsT || !mu_part) return GPBO_ERR_ARG;
    if (Mc < 1 || N < 1 || Np < N || Np % 128 || ldk % GPBO_CHUNK_GRANULE || Mc > ldk || d < 1 || d > GPBO_MAX_D)
        return GPBO_ERR_ARG;
    if (store_rows < 0 || store_rows > Np || store_rows % KS) return GPBO_ERR_ARG;
    IscArgs ls;
    for (int k = 0; k < GPBO_MAX_D; ++k) ls.isc[k] = 0.0;
    for (int k = 0; k < d; ++k) ls.isc[k] = 1.0 / (ls_host[k] * 1.4142135623730950488);
    const int KQ = (d + 3) / 4, KP = (d + 2 + 3) / 4 * 4;
    const double gamma = 2.0 * (2.0 * (d + 2) + 8.0) * 1.1102230246251565e-16;   // 2^-53
    const double eps_acc = 2.0 * (2.0 * 64.0 + 4.0 + (double)(Np / 64) + (double)(Np / 256)) * 1.1102230246251565e-16;
    const double *Bp = reinterpret_cast<const double *>(prep_buf);
    const ObsPrep *prep = reinterpret_cast<const ObsPrep *>(reinterpret_cast<const char *>(prep_buf) +
                                                            align_up((int64_t)sizeof(double) * Np * KP_MAX, 256));
    const int64_t used = (Mc + GPBO_CHUNK_GRANULE - 1) / GPBO_CHUNK_GRANULE * GPBO_CHUNK_GRANULE;
    const int OB = gpbo_kstar_mfma_slice(Np);
    dim3 grid((unsigned)(used / 256), (unsigned)(Np / OB));
#define GPBO_KM_LAUNCH(Q)                                                                                              \
    hipLaunchKernelGGL((kstar_mu_mfma_kernel<Q>), grid, dim3(256), 0, gpbo_stream(stream), Xs, Mc, (int)d, ls, Bp, prep, alpha, \
                       (int)N, gamma, eps_acc, KsT, ldk, mu_part, (int)store_rows, OB, KP)
    switch (KQ) {
        case 1: GPBO_KM_LAUNCH(1); break;
        case 2: GPBO_KM_LAUNCH(2); break;
        case 3: GPBO_KM_LAUNCH(3); break;
        case 4: GPBO_KM_LAUNCH(4); break;
        default: return GPBO_ERR_ARG;
    }
#undef GPBO_KM_LAUNCH
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}
