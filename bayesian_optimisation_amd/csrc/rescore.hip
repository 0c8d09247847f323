// Exact arg-max behind an fp32 screen (BASELINE.json config 4; the tie / first-index rule of point_selector.py:204-207).
//
// The fp32 pass (posterior_f32.hip) leaves, for every candidate c, the mean mu_c - the fp64 path's value bit for
// bit - and var32_c = prior_var - |v_c|^2 with the N^2 product done in fp32.  If the fp64 variance lies within
// tau of it,  |var64_c - var32_c| <= tau,  then with  s_lo = sqrt(max(|var32| - tau, 0)),  s_hi = sqrt(|var32| + tau)
// (both acquisitions increase with sigma: d LCB / d sigma = explore > 0, d EI / d sigma = pdf(z) > 0)
//     acq(mu_c, s_lo)  <=  acq64_c  <=  acq(mu_c, s_hi)
// and a candidate whose upper bound is below  L = max_c acq(mu_c, s_lo)  cannot be the fp64 maximum.  Everything else
// (plus every stride-th candidate, as a sample of the non-survivors) is gathered and re-scored through the fp64
// kernels; the reported maximum and its LOWEST index are taken over those fp64 values alone.
// tau is not proven a priori - a worst-case fp32 bound over N = 8192 terms is orders of magnitude above the errors
// that occur - it is checked on every call: the largest |var64 - var32| seen on the re-scored set must stay below
// tau / 4, otherwise tau is raised to 8x that error and the selection repeated.  When the survivors do not fit `cap`
// (e.g. thousands of exact ties far from the data) the caller is told to run the plain fp64 pass (stats->fallback).
#include "gpbo_internal.h"

#include <limits>

namespace {

constexpr int SB = 256;  // threads per workgroup of the streaming kernels

__device__ __forceinline__ void bounds(int kind, double mu, double var, double tau, double p0, double p1, double &lo,
                                       double &hi) {
    const double a = fabs(var);
    const double s_lo = sqrt(fmax(a - tau, 0.0)), s_hi = sqrt(a + tau);
    const double a_lo = gpbo_acquisition(kind, mu, s_lo, p0, p1);
    const double a_hi = gpbo_acquisition(kind, mu, s_hi, p0, p1);
    // both acquisitions are monotone in sigma (LCB: increasing for explore > 0, decreasing for a negative explore)
    lo = (a_hi < a_lo) ? a_hi : a_lo;   // NaN (mu NaN) stays NaN in both
    hi = (a_hi < a_lo) ? a_lo : a_hi;
}

// per-workgroup maximum of the lower bounds (NaNs skipped: they survive the selection unconditionally)
__global__ __launch_bounds__(SB) void screen_lo_kernel(const double *__restrict__ mu, const double *__restrict__ var,
                                                       int64_t M, double tau, int kind, double p0, double p1,
                                                       double *__restrict__ part) {
    __shared__ double s_val[SB / 64];
    double best = -std::numeric_limits<double>::infinity();
    for (int64_t c = (int64_t)blockIdx.x * SB + threadIdx.x; c < M; c += (int64_t)gridDim.x * SB) {
        double lo, hi;
        bounds(kind, mu[c], var[c], tau, p0, p1, lo, hi);
        if (lo > best) best = lo;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) best = fmax(best, __shfl_xor(best, off));
    if ((threadIdx.x & 63) == 0) s_val[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SB / 64; ++w) best = fmax(best, s_val[w]);
        part[blockIdx.x] = best;
    }
}

__global__ __launch_bounds__(SB) void screen_max_kernel(const double *__restrict__ part, int n, double *__restrict__ L,
                                                        unsigned long long *__restrict__ count) {
    __shared__ double s_val[SB / 64];
    double best = -std::numeric_limits<double>::infinity();
    for (int i = threadIdx.x; i < n; i += SB) best = fmax(best, part[i]);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) best = fmax(best, __shfl_xor(best, off));
    if ((threadIdx.x & 63) == 0) s_val[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SB / 64; ++w) best = fmax(best, s_val[w]);
        *L = best;
        *count = 0ull;
    }
}

// survivors: upper bound >= L, or NaN anywhere (the fp64 pass must see and count it), or one of the strided sample.
// The list order depends on the atomics; the fp64 results do not (every candidate's row is computed on its own).
__global__ __launch_bounds__(SB) void screen_select_kernel(const double *__restrict__ mu, const double *__restrict__ var,
                                                           int64_t M, double tau, int kind, double p0, double p1,
                                                           const double *__restrict__ L, int64_t stride,
                                                           int64_t *__restrict__ list, int64_t cap,
                                                           unsigned long long *__restrict__ count) {
    const double thr = *L;
    const int lane = threadIdx.x & 63;
    const int64_t step = (int64_t)gridDim.x * SB;
    const int64_t cmax = (M + step - 1) / step * step;  // whole waves stay in the loop together (ballot)
    for (int64_t c = (int64_t)blockIdx.x * SB + threadIdx.x; c < cmax; c += step) {
        bool keep = false;
        if (c < M) {
            double lo, hi;
            bounds(kind, mu[c], var[c], tau, p0, p1, lo, hi);
            keep = !(hi < thr) || (c % stride) == 0;  // !(hi < thr): also true for NaN
        }
        const unsigned long long m = __ballot(keep);
        if (m) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(count, (unsigned long long)__popcll(m));
            base = __shfl(base, 0);
            if (keep) {
                const unsigned long long pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if ((int64_t)pos < cap) list[pos] = c;
            }
        }
    }
}

// Prefix-bound screen: survivors of a threshold on the UPPER bounds ub_c >= acq64_c (gpbo_posterior_prefix_f64), plus
// every NaN (the fp64 pass must see and count it) and the strided sample.  `slack` absorbs the rounding of the two
// different summation orders (the bound comes from the fused kernel, the exact value from the column-split launch).
// One atomic per WORKGROUP: each thread counts its survivors first, the workgroup reserves a range of the list, a second
// sweep writes them (one atomic per wave with a survivor took 170-400 us on 2^21 bounds when thousands survive).
// list == nullptr: count only (the bisection probes).
__global__ __launch_bounds__(SB) void bound_select_kernel(const double *__restrict__ ub, int64_t M, double thr, double slack,
                                                          int64_t stride, int64_t *__restrict__ list, int64_t cap,
                                                          unsigned long long *__restrict__ count) {
    __shared__ unsigned s_cnt[SB];
    __shared__ unsigned long long s_base;
    const int64_t step = (int64_t)gridDim.x * SB;
    unsigned mine = 0;
    for (int64_t c = (int64_t)blockIdx.x * SB + threadIdx.x; c < M; c += step)
        mine += (!(ub[c] + slack < thr) || (c % stride) == 0) ? 1u : 0u;   // !(x < thr): also true for NaN
    s_cnt[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int t = 0; t < SB; ++t) { const unsigned v = s_cnt[t]; s_cnt[t] = tot; tot += v; }   // exclusive scan
        s_base = tot ? atomicAdd(count, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    if (!list || !mine) return;
    unsigned long long pos = s_base + s_cnt[threadIdx.x];
    for (int64_t c = (int64_t)blockIdx.x * SB + threadIdx.x; c < M; c += step)
        if (!(ub[c] + slack < thr) || (c % stride) == 0) {
            if ((int64_t)pos < cap) list[pos] = c;
            ++pos;
        }
}

// largest and smallest finite-or-infinite bound (NaNs skipped): part[0 .. n) maxima, part[n .. 2n) minima per workgroup
__global__ __launch_bounds__(SB) void bound_minmax_kernel(const double *__restrict__ ub, int64_t M, double *__restrict__ part) {
    __shared__ double s_hi[SB / 64], s_lo[SB / 64];
    double hi = -std::numeric_limits<double>::infinity(), lo = std::numeric_limits<double>::infinity();
    for (int64_t c = (int64_t)blockIdx.x * SB + threadIdx.x; c < M; c += (int64_t)gridDim.x * SB) {
        const double v = ub[c];
        if (v > hi) hi = v;
        if (v < lo) lo = v;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        hi = fmax(hi, __shfl_xor(hi, off));
        lo = fmin(lo, __shfl_xor(lo, off));
    }
    if ((threadIdx.x & 63) == 0) { s_hi[threadIdx.x >> 6] = hi; s_lo[threadIdx.x >> 6] = lo; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SB / 64; ++w) { hi = fmax(hi, s_hi[w]); lo = fmin(lo, s_lo[w]); }
        part[blockIdx.x] = hi;
        part[gridDim.x + blockIdx.x] = lo;
    }
}

__global__ __launch_bounds__(SB) void bound_minmax_finish_kernel(const double *__restrict__ part, int n, double *__restrict__ out2) {
    __shared__ double s_hi[SB / 64], s_lo[SB / 64];
    double hi = -std::numeric_limits<double>::infinity(), lo = std::numeric_limits<double>::infinity();
    for (int i = threadIdx.x; i < n; i += SB) { hi = fmax(hi, part[i]); lo = fmin(lo, part[n + i]); }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        hi = fmax(hi, __shfl_xor(hi, off));
        lo = fmin(lo, __shfl_xor(lo, off));
    }
    if ((threadIdx.x & 63) == 0) { s_hi[threadIdx.x >> 6] = hi; s_lo[threadIdx.x >> 6] = lo; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SB / 64; ++w) { hi = fmax(hi, s_hi[w]); lo = fmin(lo, s_lo[w]); }
        out2[0] = hi;
        out2[1] = lo;
    }
}

// second level of the prefix bound: of the K listed candidates keep those whose TIGHTER bound (a longer prefix) still reaches
// the threshold; the kept ones keep their order (ascending original index is not needed: the arg-max is taken by index)
__global__ __launch_bounds__(SB) void bound_refine_kernel(const double *__restrict__ ub2, const int64_t *__restrict__ list,
                                                          int64_t K, double thr, double slack, int64_t *__restrict__ list_out,
                                                          unsigned long long *__restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int64_t step = (int64_t)gridDim.x * SB;
    const int64_t imax = (K + step - 1) / step * step;
    for (int64_t i = (int64_t)blockIdx.x * SB + threadIdx.x; i < imax; i += step) {
        const bool keep = i < K && !(ub2[i] + slack < thr);
        const unsigned long long m = __ballot(keep);
        if (m) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(count, (unsigned long long)__popcll(m));
            base = __shfl(base, 0);
            if (keep) list_out[base + __popcll(m & ((1ull << lane) - 1ull))] = list[i];
        }
    }
}

__global__ __launch_bounds__(SB) void gather_rows_kernel(const double *__restrict__ Xs, int d, const int64_t *__restrict__ list,
                                                         int64_t K, double *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * SB + threadIdx.x;
    if (e >= K * d) return;
    const int64_t i = e / d;
    out[e] = Xs[list[i] * d + (e - i * d)];
}

struct RescoreOut {
    gpbo_result res;
    double err_max;  // max over the re-scored set of | sigma64^2 - |var32| |
    double pad;
};

// one workgroup: fp64 arg-max over the re-scored candidates (ties to the lowest ORIGINAL index), NaN count, and the
// largest deviation of the screen's variance from the fp64 one
__global__ __launch_bounds__(SB) void rescore_finish_kernel(const double *__restrict__ acq, const double *__restrict__ sigma,
                                                            const int64_t *__restrict__ list, int64_t K,
                                                            const double *__restrict__ var32, int64_t idx_offset,
                                                            RescoreOut *__restrict__ out) {
    __shared__ double s_val[SB / 64], s_err[SB / 64];
    __shared__ int64_t s_idx[SB / 64];
    __shared__ unsigned long long s_nan[SB / 64];
    double bv = -std::numeric_limits<double>::infinity(), err = 0.0;
    int64_t bi = std::numeric_limits<int64_t>::max();
    unsigned long long nans = 0;
    for (int64_t i = threadIdx.x; i < K; i += SB) {
        const double a = acq[i], sg = sigma[i];
        const int64_t c = list[i];
        if (a != a) ++nans;
        else if (gpbo_better(a, idx_offset + c, bv, bi)) { bv = a; bi = idx_offset + c; }
        if (var32) {
            const double e = fabs(sg * sg - fabs(var32[c]));
            if (e > err) err = e;  // NaN never raises err; NaNs are reported through the count
        }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double ov = __shfl_xor(bv, off);
        const int64_t oi = __shfl_xor(bi, off);
        err = fmax(err, __shfl_xor(err, off));
        nans += __shfl_xor(nans, off);
        if (gpbo_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_val[w] = bv; s_idx[w] = bi; s_err[w] = err; s_nan[w] = nans; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < SB / 64; ++q) {
            if (gpbo_better(s_val[q], s_idx[q], bv, bi)) { bv = s_val[q]; bi = s_idx[q]; }
            err = fmax(err, s_err[q]);
            nans += s_nan[q];
        }
        out->res.best_val = bv;
        out->res.best_idx = bi;
        out->res.nan_count = (int64_t)nans;
        out->res.reserved = 0;
        out->err_max = err;
        out->pad = 0.0;
    }
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct RescoreLayout {
    int64_t part_off, L_off, count_off, list_off, rows_off, mu_off, sig_off, acq_off, out_off, post_off, post_bytes, total;
};

constexpr int SCREEN_BLOCKS = 1024;

RescoreLayout rescore_layout(int64_t Np, int64_t cap, int64_t chunk64) {
    RescoreLayout L;
    int64_t off = 0;
    L.part_off = off; off += align_up((int64_t)sizeof(double) * SCREEN_BLOCKS, 256);
    L.L_off = off; off += 256;
    L.count_off = off; off += 256;
    L.out_off = off; off += 256;
    L.list_off = off; off += align_up((int64_t)sizeof(int64_t) * cap, 256);
    L.rows_off = off; off += align_up((int64_t)sizeof(double) * cap * GPBO_MAX_D, 256);
    L.mu_off = off; off += align_up((int64_t)sizeof(double) * cap, 256);
    L.sig_off = off; off += align_up((int64_t)sizeof(double) * cap, 256);
    L.acq_off = off; off += align_up((int64_t)sizeof(double) * cap, 256);
    L.post_off = off;
    L.post_bytes = gpbo_posterior_workspace_bytes_split(Np, chunk64, cap, GPBO_RESCORE_SPLIT_MAX);
    off += align_up(L.post_bytes, 256);
    L.total = off;
    return L;
}

}  // namespace

extern "C" int64_t gpbo_rescore_workspace_bytes(int64_t Np, int64_t cap, int64_t chunk64) {
    if (cap < 1 || gpbo_posterior_workspace_bytes(Np, chunk64, cap) < 0) return GPBO_ERR_ARG;
    return rescore_layout(Np, cap, chunk64).total;
}

extern "C" int gpbo_rescore_f64(const double *Xs, int64_t M, const double *mu, const double *var32, const double *X,
                                int64_t N, int64_t Np, int32_t d, const double *ls_host, const double *U,
                                const double *alpha, double prior_var, int32_t acq_kind, double p0, double p1,
                                int64_t idx_offset, double tau0, int64_t sample_stride, int64_t cap, int64_t chunk64,
                                gpbo_result *result, gpbo_screen_stats *stats_host, void *work, int64_t work_bytes,
                                void *stream) {
    if (!Xs || !mu || !var32 || !X || !U || !alpha || !result || !stats_host || !work) return GPBO_ERR_ARG;
    if (M < 1 || N < 1 || Np != gpbo_padded_n(N) || d < 1 || d > GPBO_MAX_D || cap < 1 || !(tau0 > 0.0) || sample_stride < 1)
        return GPBO_ERR_ARG;
    if (acq_kind != GPBO_ACQ_LCB && acq_kind != GPBO_ACQ_EI) return GPBO_ERR_ARG;
    if (gpbo_posterior_workspace_bytes(Np, chunk64, cap) < 0 || ((uintptr_t)work & 255)) return GPBO_ERR_ARG;
    const RescoreLayout L = rescore_layout(Np, cap, chunk64);
    if (work_bytes < L.total) return GPBO_ERR_WORKSPACE;
    hipStream_t st = gpbo_stream(stream);
    char *w = reinterpret_cast<char *>(work);
    double *part = reinterpret_cast<double *>(w + L.part_off);
    double *Ldev = reinterpret_cast<double *>(w + L.L_off);
    unsigned long long *count = reinterpret_cast<unsigned long long *>(w + L.count_off);
    RescoreOut *out = reinterpret_cast<RescoreOut *>(w + L.out_off);
    int64_t *list = reinterpret_cast<int64_t *>(w + L.list_off);
    double *rows = reinterpret_cast<double *>(w + L.rows_off);
    double *mu64 = reinterpret_cast<double *>(w + L.mu_off);
    double *sig64 = reinterpret_cast<double *>(w + L.sig_off);
    double *acq64 = reinterpret_cast<double *>(w + L.acq_off);
    void *post = w + L.post_off;

    int64_t nblk = (M + SB - 1) / SB;
    if (nblk > SCREEN_BLOCKS) nblk = SCREEN_BLOCKS;
    double tau = tau0;
    gpbo_screen_stats stt = {0, 0, 0, 0, tau0, 0.0};
    for (int round = 0; round < 4; ++round) {
        stt.rounds = round + 1;
        stt.tau = tau;
        hipLaunchKernelGGL(screen_lo_kernel, dim3((unsigned)nblk), dim3(SB), 0, st, mu, var32, M, tau, (int)acq_kind, p0, p1,
                           part);
        hipLaunchKernelGGL(screen_max_kernel, dim3(1), dim3(SB), 0, st, part, (int)nblk, Ldev, count);
        hipLaunchKernelGGL(screen_select_kernel, dim3((unsigned)nblk), dim3(SB), 0, st, mu, var32, M, tau, (int)acq_kind, p0,
                           p1, Ldev, sample_stride, list, cap, count);
        GPBO_CHECK_LAUNCH();
        unsigned long long K = 0;
        if (hipMemcpyAsync(&K, count, sizeof(K), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return GPBO_ERR_LAUNCH;
        stt.survivors = (int64_t)K;
        if ((int64_t)K > cap) {  // too many candidates could still be the maximum: the plain fp64 pass decides
            stt.fallback = 1;
            *stats_host = stt;
            return GPBO_OK;
        }
        const int64_t tot = (int64_t)K * d;
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((tot + SB - 1) / SB)), dim3(SB), 0, st, Xs, (int)d, list,
                           (int64_t)K, rows);
        GPBO_CHECK_LAUNCH();
        int64_t chunk = chunk64;
        const int64_t kpad = ((int64_t)K + GPBO_CHUNK_GRANULE - 1) / GPBO_CHUNK_GRANULE * GPBO_CHUNK_GRANULE;
        if (chunk > kpad) chunk = kpad;
        int rc = gpbo_posterior_acq_f64_split(rows, (int64_t)K, X, N, Np, d, ls_host, U, alpha, prior_var, acq_kind, p0, p1,
                                              0.0, 0, chunk, mu64, sig64, acq64, &out->res, post, L.post_bytes, nullptr,
                                              GPBO_RESCORE_SPLIT_MAX, 0, stream);
        if (rc != GPBO_OK) return rc;
        hipLaunchKernelGGL(rescore_finish_kernel, dim3(1), dim3(SB), 0, st, acq64, sig64, list, (int64_t)K, var32, idx_offset,
                           out);
        GPBO_CHECK_LAUNCH();
        RescoreOut h;
        if (hipMemcpyAsync(&h, out, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return GPBO_ERR_LAUNCH;
        stt.rescored += (int64_t)K;
        stt.err_max = h.err_max;
        if (4.0 * h.err_max <= tau) {  // the screen's assumption held with a factor 4 to spare on this call's sample
            if (hipMemcpyAsync(result, &out->res, sizeof(gpbo_result), hipMemcpyDeviceToDevice, st) != hipSuccess)
                return GPBO_ERR_LAUNCH;
            *stats_host = stt;
            return GPBO_OK;
        }
        tau = fmax(8.0 * h.err_max, 4.0 * tau);
    }
    stt.fallback = 1;
    *stats_host = stt;
    return GPBO_OK;
}

// Exact arg-max behind the prefix bound (the first pass: gpbo_posterior_prefix_f64, which leaves ub_c >= acq64_c for every
// candidate).  Branch and bound, everything in fp64:
//   round 0   the strided sample (and every NaN) goes through the fp64 kernels: its best value is a LOWER bound of the maximum;
//   round r   every candidate whose upper bound reaches the best exact value seen so far survives; if the survivors fit `cap`
//             they (and the sample) are re-scored and the maximum and its LOWEST index are taken over those fp64 values - a
//             pruned candidate has acq64 <= ub < an exact value, so it is neither the maximum nor tied with it; otherwise the
//             first `refine` survivors are re-scored to raise the threshold and the selection is repeated (at most 3 times).
// When the survivors still do not fit (a flat mean, thousands of ties, an exploration weight that dwarfs the mean) the caller
// is told to run the plain pass (stats->fallback).  stats: tau = the last threshold, err_max unused.
static int bound_select_impl(const double *Xs, int64_t M, const double *ub, const double *X, int64_t N, int64_t Np,
                             int32_t d, const double *ls_host, const double *U, const double *alpha,
                             double prior_var, int32_t acq_kind, double p0, double p1, int64_t idx_offset,
                             int64_t sample_stride, int64_t cap, int64_t chunk64, int64_t n_prefix2,
                             gpbo_result *result, gpbo_screen_stats *stats_host, void *work, int64_t work_bytes,
                             void *stream) {
    if (!Xs || !ub || !X || !U || !alpha || !result || !stats_host || !work) return GPBO_ERR_ARG;
    if (n_prefix2 < 0 || n_prefix2 > Np || n_prefix2 % 128) return GPBO_ERR_ARG;
    if (M < 1 || N < 1 || Np != gpbo_padded_n(N) || d < 1 || d > GPBO_MAX_D || cap < 1 || sample_stride < 1) return GPBO_ERR_ARG;
    if (acq_kind != GPBO_ACQ_LCB && acq_kind != GPBO_ACQ_EI) return GPBO_ERR_ARG;
    if (acq_kind == GPBO_ACQ_LCB && !(p0 >= 0.0)) return GPBO_ERR_ARG;   // the bound needs an acquisition that increases with sigma
    if (gpbo_posterior_workspace_bytes(Np, chunk64, cap) < 0 || ((uintptr_t)work & 255)) return GPBO_ERR_ARG;
    const RescoreLayout L = rescore_layout(Np, cap, chunk64);
    if (work_bytes < L.total) return GPBO_ERR_WORKSPACE;
    hipStream_t st = gpbo_stream(stream);
    char *w = reinterpret_cast<char *>(work);
    unsigned long long *count = reinterpret_cast<unsigned long long *>(w + L.count_off);
    RescoreOut *out = reinterpret_cast<RescoreOut *>(w + L.out_off);
    int64_t *list = reinterpret_cast<int64_t *>(w + L.list_off);
    double *rows = reinterpret_cast<double *>(w + L.rows_off);
    double *mu64 = reinterpret_cast<double *>(w + L.mu_off);
    double *sig64 = reinterpret_cast<double *>(w + L.sig_off);
    double *acq64 = reinterpret_cast<double *>(w + L.acq_off);
    void *post = w + L.post_off;
    int64_t nblk = (M + SB - 1) / SB;
    if (nblk > SCREEN_BLOCKS) nblk = SCREEN_BLOCKS;
    const double inf = std::numeric_limits<double>::infinity();
    gpbo_screen_stats stt = {0, 0, 0, 0, 0.0, 0.0};

    // exact fp64 values of the first K entries of `list`; their arg-max (lowest original index) lands in *out
    auto exact = [&](int64_t K, RescoreOut *h) -> int {
        const int64_t tot = K * d;
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((tot + SB - 1) / SB)), dim3(SB), 0, st, Xs, (int)d, list, K, rows);
        GPBO_CHECK_LAUNCH();
        int64_t chunk = chunk64;
        const int64_t kpad = (K + GPBO_CHUNK_GRANULE - 1) / GPBO_CHUNK_GRANULE * GPBO_CHUNK_GRANULE;
        if (chunk > kpad) chunk = kpad;
        int rc = gpbo_posterior_acq_f64_split(rows, K, X, N, Np, d, ls_host, U, alpha, prior_var, acq_kind, p0, p1, 0.0, 0,
                                              chunk, mu64, sig64, acq64, &out->res, post, L.post_bytes, nullptr,
                                              GPBO_RESCORE_SPLIT_MAX, 0, stream);
        if (rc != GPBO_OK) return rc;
        hipLaunchKernelGGL(rescore_finish_kernel, dim3(1), dim3(SB), 0, st, acq64, sig64, list, K, (const double *)nullptr,
                           idx_offset, out);
        GPBO_CHECK_LAUNCH();
        if (hipMemcpyAsync(h, out, sizeof(*h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return GPBO_ERR_LAUNCH;
        stt.rescored += K;
        ++stt.rounds;   // = launches of the fp64 kernels on gathered rows
        return GPBO_OK;
    };
    const int64_t no_sample = std::numeric_limits<int64_t>::max();
    auto select = [&](double thr, int64_t stride, unsigned long long *K, bool count_only = false) -> int {
        if (hipMemsetAsync(count, 0, sizeof(unsigned long long), st) != hipSuccess) return GPBO_ERR_LAUNCH;
        const double slack = (thr == inf) ? 0.0 : 1e-10 * fmax(1.0, fabs(thr));
        hipLaunchKernelGGL(bound_select_kernel, dim3((unsigned)nblk), dim3(SB), 0, st, ub, M, thr, slack, stride,
                           count_only ? (int64_t *)nullptr : list, cap, count);
        GPBO_CHECK_LAUNCH();
        if (hipMemcpyAsync(K, count, sizeof(*K), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return GPBO_ERR_LAUNCH;
        return GPBO_OK;
    };

    RescoreOut h;
    unsigned long long K = 0;
    int rc;
    double thr = -inf;
    double *part = reinterpret_cast<double *>(w + L.part_off);
    double *Ldev = reinterpret_cast<double *>(w + L.L_off);
    // smallest and largest value of n bounds (NaNs skipped)
    auto minmax = [&](const double *v, int64_t n, double *mm) -> int {
        int64_t blk = (n + SB - 1) / SB;
        if (blk > SCREEN_BLOCKS / 2) blk = SCREEN_BLOCKS / 2;
        hipLaunchKernelGGL(bound_minmax_kernel, dim3((unsigned)blk), dim3(SB), 0, st, v, n, part);
        hipLaunchKernelGGL(bound_minmax_finish_kernel, dim3(1), dim3(SB), 0, st, part, (int)blk, Ldev);
        GPBO_CHECK_LAUNCH();
        if (hipMemcpyAsync(mm, Ldev, 2 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return GPBO_ERR_LAUNCH;
        return GPBO_OK;
    };
    // a level between mm[1] and mm[0] that `counter(level)` candidates reach, want_lo <= count <= want_hi
    auto bisect = [&](const double *mm, int64_t want_lo, int64_t want_hi, auto &&counter, double *level) -> int {
        *level = inf;   // "none found"
        if (!(mm[0] > mm[1]) || !(mm[0] < inf) || !(mm[1] > -inf)) return GPBO_OK;
        double lo = mm[1], hi = mm[0];
        for (int it = 0; it < 30; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (!(mid > lo) || !(mid < hi)) break;
            unsigned long long c = 0;
            int r = counter(mid, &c);
            if (r != GPBO_OK) return r;
            if ((int64_t)c > want_hi) lo = mid;
            else if ((int64_t)c < want_lo) hi = mid;
            else { *level = mid; break; }
        }
        return GPBO_OK;
    };
    auto count_ub = [&](double level, unsigned long long *c) -> int { return select(level, no_sample, c, true); };
    const int64_t want_hi = cap < 4096 ? cap : 4096, want_lo = want_hi / 8;
    double mm[2] = {0.0, 0.0};
    rc = minmax(ub, M, mm);
    if (rc != GPBO_OK) return rc;

    // ---- the usual case in one sweep: second-level bounds BEFORE any fp64 row, one fp64 launch --------------------------
    // The K1 candidates with the largest first-level bounds (a few times more than can survive) get the second-level
    // bound; the few hundred with the largest SECOND-level bounds are re-scored: t = their best exact value.  If t is at
    // least both levels, nobody else can reach it: outside the K1, ub1 < level1 <= t; inside, ub2 < level2 <= t.  Done.
    // Anything else (levels not found, t below a level) leaves t as a valid threshold for the general loop below.
    if (n_prefix2 > 0) {
        int64_t k1 = M / 64;
        if (k1 < 2048) k1 = 2048;
        if (k1 > cap / 2) k1 = cap / 2;
        double level1 = inf;
        rc = bisect(mm, k1 - k1 / 3, k1 + k1 / 2, count_ub, &level1);
        if (rc != GPBO_OK) return rc;
        if (level1 < inf) {
            rc = select(level1, no_sample, &K);
            if (rc != GPBO_OK) return rc;
            const int64_t K1 = (int64_t)K;
            if (K1 >= 1 && K1 <= cap) {
                const int64_t tot = K1 * d;
                hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((tot + SB - 1) / SB)), dim3(SB), 0, st, Xs, (int)d, list, K1,
                                   rows);
                GPBO_CHECK_LAUNCH();
                int64_t chunk = chunk64;
                const int64_t kpad = (K1 + GPBO_CHUNK_GRANULE - 1) / GPBO_CHUNK_GRANULE * GPBO_CHUNK_GRANULE;
                if (chunk > kpad) chunk = kpad;
                double *ub2 = sig64;   // (acq64 / mu64 are overwritten by the fp64 launch below; sig64 is too, AFTER ub2's last use)
                rc = gpbo_posterior_acq_f64_split(rows, K1, X, N, Np, d, ls_host, U, alpha, prior_var, acq_kind, p0, p1, 0.0, 0,
                                                  chunk, nullptr, nullptr, ub2, &out->res, post, L.post_bytes, nullptr,
                                                  GPBO_RESCORE_SPLIT_MAX, n_prefix2, stream);
                if (rc != GPBO_OK) return rc;
                int64_t *list2 = reinterpret_cast<int64_t *>(mu64);
                int64_t rblk = (K1 + SB - 1) / SB;
                if (rblk > SCREEN_BLOCKS) rblk = SCREEN_BLOCKS;
                auto refine_ub2 = [&](double level, unsigned long long *c) -> int {   // list2 <- {i : ub2_i >= level}
                    if (hipMemsetAsync(count, 0, sizeof(unsigned long long), st) != hipSuccess) return GPBO_ERR_LAUNCH;
                    hipLaunchKernelGGL(bound_refine_kernel, dim3((unsigned)rblk), dim3(SB), 0, st, ub2, list, K1, level,
                                       1e-10 * fmax(1.0, fabs(level)), list2, count);
                    GPBO_CHECK_LAUNCH();
                    if (hipMemcpyAsync(c, count, sizeof(*c), hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess)
                        return GPBO_ERR_LAUNCH;
                    return GPBO_OK;
                };
                double mm2[2], level2 = inf;
                rc = minmax(ub2, K1, mm2);
                if (rc != GPBO_OK) return rc;
                const int64_t t_hi = K1 < 1024 ? K1 : 1024;
                rc = bisect(mm2, t_hi / 4, t_hi, refine_ub2, &level2);
                if (rc != GPBO_OK) return rc;
                if (level2 < inf) {
                    unsigned long long KT = 0;
                    rc = refine_ub2(level2, &KT);   // (the probe that found the level left exactly this list; kept explicit)
                    if (rc != GPBO_OK) return rc;
                    if (KT >= 1 && (int64_t)KT <= cap) {
                        // (the K1 list is not needed again: a miss goes to the general loop, which selects afresh)
                        if (hipMemcpyAsync(list, list2, sizeof(int64_t) * KT, hipMemcpyDeviceToDevice, st) != hipSuccess)
                            return GPBO_ERR_LAUNCH;
                        rc = exact((int64_t)KT, &h);
                        if (rc != GPBO_OK) return rc;
                        thr = h.res.best_val;
                        stt.survivors = K1;
                        stt.tau = thr;
                        if (thr >= level1 && thr >= level2) {
                            if (hipMemcpyAsync(result, &out->res, sizeof(gpbo_result), hipMemcpyDeviceToDevice, st) != hipSuccess)
                                return GPBO_ERR_LAUNCH;
                            *stats_host = stt;
                            return GPBO_OK;
                        }
                    }
                }
            }
        }
    }

    // ---- general route ---------------------------------------------------------------------------------------------------
    // A lower bound of the maximum.  The candidates with the LARGEST bounds are the likely winners: bisect for a level that
    // keeps a few thousand of them (every select is one pass over the bounds), re-score those: their best exact value is
    // the threshold - on the benchmark problem it IS the maximum.
    if (!(thr > -inf)) {
        double level = inf;
        rc = bisect(mm, want_lo, want_hi, count_ub, &level);
        if (rc != GPBO_OK) return rc;
        if (level < inf) {   // list the candidates of this level (NaN bounds included) and re-score them
            rc = select(level, no_sample, &K);
            if (rc != GPBO_OK) return rc;
            if ((int64_t)K >= 1 && (int64_t)K <= cap) {
                rc = exact((int64_t)K, &h);
                if (rc != GPBO_OK) return rc;
                thr = h.res.best_val;
            }
        }
    }
    if (!(thr > -inf)) {
        // no such level (e.g. thousands of equal bounds): the strided sample (and every NaN) gives the first threshold
        rc = select(inf, sample_stride, &K);
        if (rc != GPBO_OK) return rc;
        stt.survivors = (int64_t)K;
        if ((int64_t)K > cap) { stt.fallback = 1; *stats_host = stt; return GPBO_OK; }
        rc = exact((int64_t)K, &h);
        if (rc != GPBO_OK) return rc;
        thr = h.res.best_val;   // -inf when every sampled acquisition is NaN: everything survives
    }
    const int64_t refine = cap < 4096 ? cap : 4096;
    for (int round = 1; round <= 4; ++round) {
        stt.tau = thr;
        rc = select(thr, no_sample, &K);
        if (rc != GPBO_OK) return rc;
        stt.survivors = (int64_t)K;
        if ((int64_t)K <= cap) {
            if (n_prefix2 > 0 && (int64_t)K > 4 * refine) {
                // second level: a longer prefix for the survivors only (tighter bounds at (n_prefix2 / Np)^2 of the
                // exact cost); what still reaches the threshold goes to the fp64 kernels
                const int64_t tot = (int64_t)K * d;
                hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((tot + SB - 1) / SB)), dim3(SB), 0, st, Xs, (int)d, list,
                                   (int64_t)K, rows);
                GPBO_CHECK_LAUNCH();
                int64_t chunk = chunk64;
                const int64_t kpad = ((int64_t)K + GPBO_CHUNK_GRANULE - 1) / GPBO_CHUNK_GRANULE * GPBO_CHUNK_GRANULE;
                if (chunk > kpad) chunk = kpad;
                rc = gpbo_posterior_acq_f64_split(rows, (int64_t)K, X, N, Np, d, ls_host, U, alpha, prior_var, acq_kind, p0, p1,
                                                  0.0, 0, chunk, nullptr, nullptr, acq64, &out->res, post, L.post_bytes, nullptr,
                                                  GPBO_RESCORE_SPLIT_MAX, n_prefix2, stream);
                if (rc != GPBO_OK) return rc;
                int64_t *list2 = reinterpret_cast<int64_t *>(mu64);   // (cap x 8 bytes, not used by the call above)
                if (hipMemsetAsync(count, 0, sizeof(unsigned long long), st) != hipSuccess) return GPBO_ERR_LAUNCH;
                int64_t rblk = ((int64_t)K + SB - 1) / SB;
                if (rblk > SCREEN_BLOCKS) rblk = SCREEN_BLOCKS;
                hipLaunchKernelGGL(bound_refine_kernel, dim3((unsigned)rblk), dim3(SB), 0, st, acq64, list, (int64_t)K, thr,
                                   1e-10 * fmax(1.0, fabs(thr)), list2, count);
                GPBO_CHECK_LAUNCH();
                unsigned long long K2 = 0;
                if (hipMemcpyAsync(&K2, count, sizeof(K2), hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipStreamSynchronize(st) != hipSuccess)
                    return GPBO_ERR_LAUNCH;
                if (K2 > K) return GPBO_ERR_LAUNCH;
                if (hipMemcpyAsync(list, list2, sizeof(int64_t) * K2, hipMemcpyDeviceToDevice, st) != hipSuccess)
                    return GPBO_ERR_LAUNCH;
                K = K2;   // (never empty: the candidate that set the threshold has ub2 >= its exact value)
                if (K == 0) return GPBO_ERR_LAUNCH;
            }
            // the fp64 kernels on a list run in small chunks: beyond an eighth of the candidates the plain pass is cheaper
            if ((int64_t)K > M / 8 && (int64_t)K > 4 * refine) break;
            rc = exact((int64_t)K, &h);
            if (rc != GPBO_OK) return rc;
            if (hipMemcpyAsync(result, &out->res, sizeof(gpbo_result), hipMemcpyDeviceToDevice, st) != hipSuccess)
                return GPBO_ERR_LAUNCH;
            *stats_host = stt;
            return GPBO_OK;
        }
        if (round == 4) break;
        rc = exact(refine, &h);   // some of the survivors: a better lower bound of the maximum
        if (rc != GPBO_OK) return rc;
        if (!(h.res.best_val > thr)) break;   // no progress: the bound does not separate these candidates
        thr = h.res.best_val;
    }
    stt.fallback = 1;
    *stats_host = stt;
    return GPBO_OK;
}

extern "C" int gpbo_bound_select_f64(const double *Xs, int64_t M, const double *ub, const double *X, int64_t N, int64_t Np,
                                     int32_t d, const double *ls_host, const double *U, const double *alpha,
                                     double prior_var, int32_t acq_kind, double p0, double p1, int64_t idx_offset,
                                     int64_t sample_stride, int64_t cap, int64_t chunk64, int64_t n_prefix2,
                                     gpbo_result *result, gpbo_screen_stats *stats_host, void *work, int64_t work_bytes,
                                     void *stream) {
    return bound_select_impl(Xs, M, ub, X, N, Np, d, ls_host, U, alpha, prior_var, acq_kind, p0, p1, idx_offset, sample_stride,
                             cap, chunk64, n_prefix2, result, stats_host, work, work_bytes, stream);
}
