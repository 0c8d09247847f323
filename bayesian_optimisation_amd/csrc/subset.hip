// Order-independent pruning for the exact prefix bound: a farthest-point ORDER of the observations
// (VERDICT round 3 item 1).
//
// The bound of sigma_acq.hip / rescore.hip uses the variance reduction from the FIRST J observations of the factorised
// problem:  sigma_c^2 = c - |v_c|^2 <= c - |v_c[:J]|^2, v_c = U^T k_c (U triangular: component j depends on observations
// 1..j only; /root/reference/point_selector.py:91).  Which observations come first decides how much the bound prunes: a
// Sobol stream covers the domain with any prefix, a sorted or clustered history does not.  This file computes a permutation
// of the observations - J members by farthest-point sampling in length-scale units (each new member is the observation
// farthest from the members so far; the first one is the observation farthest from the centroid; ties to the lowest
// index), then all the others in index order - and gathers X and y in that order.  The caller factorises the PERMUTED
// problem (the posterior of a GP does not depend on the order of its observations) and every pass - the plain one and the
// bound's two levels - then works on that one factorisation: the bound's |v[:J]|^2 is a partial sum of the very squares
// the plain pass adds up, for any history.  (Round 3 kept the arrival order and gave the bound a factorisation of its
// own, chol(K_SS): the same members, but two different sets of rounding errors to compare.)
#include "gpbo_internal.h"

#include <limits>

namespace {

constexpr int FT = 1024;  // threads of the single workgroup

struct FpsLs {
    double isc[GPBO_MAX_D];  // 1 / ls_k
};

__device__ __forceinline__ void block_argmax(double &v, int64_t &i, double *s_val, int64_t *s_idx) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double ov = __shfl_xor(v, off);
        const int64_t oi = __shfl_xor(i, off);
        if (gpbo_better(ov, oi, v, i)) { v = ov; i = oi; }
    }
    if (lane == 0) { s_val[w] = v; s_idx[w] = i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bv = s_val[0];
        int64_t bi = s_idx[0];
        for (int q = 1; q < FT / 64; ++q)
            if (gpbo_better(s_val[q], s_idx[q], bv, bi)) { bv = s_val[q]; bi = s_idx[q]; }
        s_val[0] = bv;
        s_idx[0] = bi;
    }
    __syncthreads();
    v = s_val[0];
    i = s_idx[0];
    __syncthreads();
}

// extension to J2 members: unchosen observations (mind > -inf) in index order; thread t of the FT owns a contiguous range
__device__ __forceinline__ void fps_extend(int64_t N, int64_t J, int64_t J2, const double *__restrict__ mind,
                                           int64_t *__restrict__ perm, long long *s_scan) {
    const int tid = threadIdx.x;
    const double ninf = -std::numeric_limits<double>::infinity();
    if (J2 > J) {
        const int64_t per = (N + FT - 1) / FT, lo = (int64_t)tid * per, hi = (lo + per < N) ? lo + per : N;
        long long cnt = 0;
        for (int64_t i = lo; i < hi; ++i) cnt += (mind[i] != ninf);
        s_scan[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < FT; off <<= 1) {  // inclusive Hillis-Steele scan
            const long long add = (tid >= off) ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += add;
            __syncthreads();
        }
        long long pos = s_scan[tid] - cnt;
        for (int64_t i = lo; i < hi; ++i) {
            if (mind[i] != ninf) {
                if (pos < J2 - J) perm[J + pos] = i;
                ++pos;
            }
        }
    }
}

// perm[0 .. J) = farthest-point sequence; perm[J .. J2) = the first J2 - J observations not among them, in index order.
// One workgroup.  PTS > 0: every thread keeps its PTS observations (scaled coordinates) and their distances to the
// member set in registers (N <= 1024 PTS) - a selection step is then d fmas per point and one block arg-max, ~1 us;
// PTS == 0: any N, coordinates re-read and distances kept in `mind` (global) every step.
template <int PTS, int D>
__global__ __launch_bounds__(FT) void fps_kernel(const double *__restrict__ X, int64_t N, int d_rt, FpsLs ls, int64_t J, int64_t J2,
                                                 double *__restrict__ mind, int64_t *__restrict__ perm) {
    __shared__ double s_val[FT / 64];
    __shared__ int64_t s_idx[FT / 64];
    __shared__ double s_c[GPBO_MAX_D];
    __shared__ long long s_scan[FT];
    const int tid = threadIdx.x;
    const int d = (PTS > 0) ? D : d_rt;
    const double ninf = -std::numeric_limits<double>::infinity();
    constexpr int NP = PTS > 0 ? PTS : 1, ND = PTS > 0 ? D : 1;
    double xr[NP][ND], md[NP];
    if (PTS > 0) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int64_t i = tid + (int64_t)FT * q;
#pragma unroll
            for (int k = 0; k < ND; ++k) xr[q][k] = (i < N) ? X[i * D + k] * ls.isc[k] : 0.0;
            md[q] = std::numeric_limits<double>::infinity();
        }
    }
    // centroid (fixed-order reduction: per-thread partial sums, then thread 0 over the 1024 partials of each coordinate)
    for (int k = 0; k < d; ++k) {
        double s = 0.0;
        for (int64_t i = tid; i < N; i += FT) s += X[i * d + k] * ls.isc[k];
        reinterpret_cast<double *>(s_scan)[tid] = s;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int q = 0; q < FT; ++q) t += reinterpret_cast<double *>(s_scan)[q];
            s_c[k] = t / (double)N;
        }
        __syncthreads();
    }
    // distance of this thread's points to s_c, folded into their running minimum; returns the thread's best (value, index)
    auto sweep = [&](int64_t member, double &bv, int64_t &bi) {
        bv = ninf;
        bi = std::numeric_limits<int64_t>::max();
        if (PTS > 0) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int64_t i = tid + (int64_t)FT * q;
                double dist = 0.0;
#pragma unroll
                for (int k = 0; k < ND; ++k) {
                    const double df = xr[q][k] - s_c[k];
                    dist = fma(df, df, dist);
                }
                double m = (member < 0) ? dist : fmin(md[q], dist);
                if (member >= 0) {
                    if (i == member) m = ninf;  // a member is never chosen again (its duplicates: distance 0, chosen last)
                    md[q] = m;
                }
                if (i < N && gpbo_better(m, i, bv, bi)) { bv = m; bi = i; }
            }
        } else {
            for (int64_t i = tid; i < N; i += FT) {
                double dist = 0.0;
                for (int k = 0; k < d; ++k) {
                    const double df = X[i * d + k] * ls.isc[k] - s_c[k];
                    dist = fma(df, df, dist);
                }
                double m = dist;
                if (member >= 0) {
                    m = fmin(mind[i], dist);
                    if (i == member) m = ninf;
                    mind[i] = m;
                } else {
                    mind[i] = std::numeric_limits<double>::infinity();
                }
                if (gpbo_better(m, i, bv, bi)) { bv = m; bi = i; }
            }
        }
    };
    double bv;
    int64_t bi;
    sweep(-1, bv, bi);  // first member: farthest from the centroid
    block_argmax(bv, bi, s_val, s_idx);
    for (int64_t j = 0; j < J; ++j) {
        const int64_t p = bi;
        if (tid == 0) perm[j] = p;
        if (tid < d) s_c[tid] = X[p * d + tid] * ls.isc[tid];
        __syncthreads();
        sweep(p, bv, bi);
        block_argmax(bv, bi, s_val, s_idx);
    }
    if (PTS > 0) {  // the extension below reads the membership from `mind`
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int64_t i = tid + (int64_t)FT * q;
            if (i < N) mind[i] = md[q];
        }
        __syncthreads();
    }
    fps_extend(N, J, J2, mind, perm, s_scan);
}

// ---- the same selection as ONE LAUNCH PER STEP, for sizes whose points do not fit one workgroup's registers --------------
// (8 points x 8 coordinates per thread of the 1024 spill: 12 us per step, 6.9 ms of sampling at N = 8192, d = 8; the
// single-workgroup global-memory form is slower still.)  Every workgroup folds the newest member into the running minimum
// of its own points and writes its best (value, index); the workgroup that finishes LAST - an atomic ticket, no waiting -
// reduces the partials in workgroup order and publishes the next member.  Same arithmetic, same tie rule (lowest index),
// hence the same sequence as fps_kernel; the cost is a kernel boundary per member (~2.5 us).
constexpr int FG = 256;
struct FpsState {
    int64_t member;        // newest member (read by the next launch)
    unsigned int ticket;   // workgroups of the current launch that have finished
    unsigned int pad;
    double centre[GPBO_MAX_D];   // scaled coordinates the next launch measures distances to
};

__global__ __launch_bounds__(FT) void fps_centroid_kernel(const double *__restrict__ X, int64_t N, int d, FpsLs ls,
                                                          FpsState *__restrict__ stt) {
    __shared__ double s_part[FT];
    const int tid = threadIdx.x;
    for (int k = 0; k < d; ++k) {   // the fixed-order reduction of fps_kernel
        double s = 0.0;
        for (int64_t i = tid; i < N; i += FT) s += X[i * d + k] * ls.isc[k];
        s_part[tid] = s;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int q = 0; q < FT; ++q) t += s_part[q];
            stt->centre[k] = t / (double)N;
        }
        __syncthreads();
    }
    if (tid == 0) { stt->member = -1; stt->ticket = 0; }
}

__global__ __launch_bounds__(FG) void fps_step_kernel(const double *__restrict__ X, int64_t N, int d, FpsLs ls,
                                                      double *__restrict__ mind, FpsState *__restrict__ stt,
                                                      double *__restrict__ pval, int64_t *__restrict__ pidx,
                                                      int64_t *__restrict__ perm, int64_t j, int64_t J) {
    __shared__ double s_val[FG / 64];
    __shared__ int64_t s_idx[FG / 64];
    __shared__ double s_c[GPBO_MAX_D];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const double ninf = -std::numeric_limits<double>::infinity();
    const int64_t member = stt->member;
    if (tid < d) s_c[tid] = stt->centre[tid];
    __syncthreads();
    double bv = ninf;
    int64_t bi = std::numeric_limits<int64_t>::max();
    for (int64_t i = (int64_t)blockIdx.x * FG + tid; i < N; i += (int64_t)gridDim.x * FG) {
        double dist = 0.0;
        for (int k = 0; k < d; ++k) {
            const double df = X[i * d + k] * ls.isc[k] - s_c[k];
            dist = fma(df, df, dist);
        }
        double m = dist;
        if (member >= 0) {
            m = fmin(mind[i], dist);
            if (i == member) m = ninf;  // a member is never chosen again
            mind[i] = m;
        } else {
            mind[i] = std::numeric_limits<double>::infinity();
        }
        if (gpbo_better(m, i, bv, bi)) { bv = m; bi = i; }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double ov = __shfl_xor(bv, off);
        const int64_t oi = __shfl_xor(bi, off);
        if (gpbo_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { s_val[w] = bv; s_idx[w] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int q = 1; q < FG / 64; ++q)
            if (gpbo_better(s_val[q], s_idx[q], bv, bi)) { bv = s_val[q]; bi = s_idx[q]; }
        // The partials travel as device-scope (sc1, write-through) atomic stores, acknowledged before the ticket is taken; the
        // last workgroup reads them with device-scope atomic loads.  No __threadfence: a release at device scope writes the
        // whole L2's dirty lines back (the running minima this launch has just stored - which only the NEXT launch reads,
        // after the kernel boundary): two such fences were most of a 7-us step.
        __hip_atomic_store(pval + blockIdx.x, bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pidx + blockIdx.x, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = (__hip_atomic_fetch_add(&stt->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (s_last) {  // every workgroup's partial has been acknowledged: reduce them (the order does not matter: largest value,
        double v = ninf;   // lowest index among equals) and publish the next member
        int64_t p = std::numeric_limits<int64_t>::max();
        for (unsigned b = tid; b < gridDim.x; b += FG) {
            const double ov = __hip_atomic_load(pval + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int64_t oi = __hip_atomic_load(pidx + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gpbo_better(ov, oi, v, p)) { v = ov; p = oi; }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double ov = __shfl_xor(v, off);
            const int64_t oi = __shfl_xor(p, off);
            if (gpbo_better(ov, oi, v, p)) { v = ov; p = oi; }
        }
        if (lane == 0) { s_val[w] = v; s_idx[w] = p; }
        __syncthreads();
        if (tid == 0) {
            for (int q = 1; q < FG / 64; ++q)
                if (gpbo_better(s_val[q], s_idx[q], v, p)) { v = s_val[q]; p = s_idx[q]; }
            if (j < J) perm[j] = p;
            stt->member = p;
            stt->ticket = 0;
        }
        __syncthreads();
        if (tid == 0) s_idx[0] = p;
        __syncthreads();
        const int64_t pw = s_idx[0];
        if (tid < d) stt->centre[tid] = X[pw * d + tid] * ls.isc[tid];
    }
}

__global__ __launch_bounds__(FT) void fps_extend_kernel(int64_t N, int64_t J, int64_t J2, const double *__restrict__ mind,
                                                        int64_t *__restrict__ perm) {
    __shared__ long long s_scan[FT];
    fps_extend(N, J, J2, mind, perm, s_scan);
}

__global__ void gather_obs_kernel(const double *__restrict__ X, int d, const int64_t *__restrict__ perm, int64_t n,
                                  double *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * d) return;
    out[e] = X[perm[e / d] * d + (e % d)];
}

struct OrderLayout {
    int64_t mind_off, fps_off, total;
};
constexpr int FPS_MAXG = 256;  // workgroups of a selection step at most

OrderLayout order_layout(int64_t N) {
    OrderLayout L;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    L.mind_off = take((int64_t)sizeof(double) * N);
    L.fps_off = take((int64_t)sizeof(FpsState) + (int64_t)FPS_MAXG * (sizeof(double) + sizeof(int64_t)) + 256);
    L.total = off;
    return L;
}

__global__ void gather_y_kernel(const double *__restrict__ y, const int64_t *__restrict__ perm, int64_t n, double *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = y[perm[e]];
}

}  // namespace

extern "C" int64_t gpbo_fps_order_workspace_bytes(int64_t N) {
    if (N < 1) return GPBO_ERR_ARG;
    return order_layout(N).total;
}

// perm_out [N] int64: perm[0 .. J) = the farthest-point sequence, perm[J .. N) = every other observation in index order.
// Xp_out [N x d] / yp_out [N] (optional; y may be NULL when yp_out is): rows perm[i] of X / y.  All device memory.
extern "C" int gpbo_fps_order_f64(const double *X, const double *y, int64_t N, int32_t d, const double *ls_host, int64_t J,
                                  int64_t *perm_out, double *Xp_out, double *yp_out, void *work, int64_t work_bytes,
                                  void *stream) {
    if (!X || !ls_host || !perm_out || !work || (yp_out && !y)) return GPBO_ERR_ARG;
    if (N < 1 || d < 1 || d > GPBO_MAX_D || J < 1 || J > N) return GPBO_ERR_ARG;
    if ((uintptr_t)work & 255) return GPBO_ERR_ARG;
    const OrderLayout L = order_layout(N);
    if (work_bytes < L.total) return GPBO_ERR_WORKSPACE;
    FpsLs ls;
    for (int k = 0; k < GPBO_MAX_D; ++k) ls.isc[k] = 0.0;
    for (int k = 0; k < d; ++k) {
        if (!(ls_host[k] > 0.0)) return GPBO_ERR_ARG;
        ls.isc[k] = 1.0 / ls_host[k];
    }
    hipStream_t st = gpbo_stream(stream);
    char *w = reinterpret_cast<char *>(work);
    double *mind = reinterpret_cast<double *>(w + L.mind_off);
    // one workgroup with every thread's observations in registers while that is the faster form (measured at N = 8192 /
    // 6000: 8 points per thread at d = 4 - 116 bytes of spills - 3.1 ms against 4.3 by launches; at d = 8 - 404 bytes -
    // 6.9 against 4.4; at d = 6 2.5 against 2.2); beyond: one launch per member (fps_step_kernel)
#define GPBO_FPS(P, DD) hipLaunchKernelGGL((fps_kernel<P, DD>), dim3(1), dim3(FT), 0, st, X, N, (int)d, ls, J, N, mind, perm_out)
    const int64_t pts = (N + FT - 1) / FT;
    bool launched = false;
#define GPBO_FPS_D(DD)                                                                   \
    if (!launched && d == DD) {                                                          \
        if (pts <= 1) { GPBO_FPS(1, DD); launched = true; }                              \
        else if (pts <= 2) { GPBO_FPS(2, DD); launched = true; }                         \
        else if (pts <= 4 && DD <= 8) { GPBO_FPS(4, DD); launched = true; }              \
        else if (pts <= 8 && DD <= 4) { GPBO_FPS(8, DD); launched = true; }              \
    }
    GPBO_FPS_D(1) GPBO_FPS_D(2) GPBO_FPS_D(3) GPBO_FPS_D(4) GPBO_FPS_D(5) GPBO_FPS_D(6) GPBO_FPS_D(7) GPBO_FPS_D(8)
    GPBO_FPS_D(9) GPBO_FPS_D(10) GPBO_FPS_D(11) GPBO_FPS_D(12) GPBO_FPS_D(13) GPBO_FPS_D(14) GPBO_FPS_D(15) GPBO_FPS_D(16)
    if (!launched) {
        // one launch per member (the state block: FpsState, then the workgroups' partial values and indices)
        FpsState *stt = reinterpret_cast<FpsState *>(w + L.fps_off);
        double *pval = reinterpret_cast<double *>(w + L.fps_off + ((sizeof(FpsState) + 255) / 256) * 256);
        int64_t *pidx = reinterpret_cast<int64_t *>(pval + FPS_MAXG);
        int64_t G = (N + FG - 1) / FG;
        if (G > FPS_MAXG) G = FPS_MAXG;
        hipLaunchKernelGGL(fps_centroid_kernel, dim3(1), dim3(FT), 0, st, X, N, (int)d, ls, stt);
        for (int64_t j = 0; j <= J; ++j)   // launch j: folds member j - 1 in and picks member j (the last one only folds)
            hipLaunchKernelGGL(fps_step_kernel, dim3((unsigned)G), dim3(FG), 0, st, X, N, (int)d, ls, mind, stt, pval, pidx, perm_out, j, J);
        hipLaunchKernelGGL(fps_extend_kernel, dim3(1), dim3(FT), 0, st, N, J, N, mind, perm_out);
    }
#undef GPBO_FPS_D
#undef GPBO_FPS
    if (Xp_out) {
        const int64_t tot = N * d;
        hipLaunchKernelGGL(gather_obs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, X, (int)d, perm_out, N, Xp_out);
    }
    if (yp_out) hipLaunchKernelGGL(gather_y_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, y, perm_out, N, yp_out);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}
