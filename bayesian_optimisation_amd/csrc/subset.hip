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

#include <cstdio>
#include <cstdlib>
#include <limits>

namespace {

constexpr int FT = 1024;  // threads of the single workgroup

struct FpsLs {
    double isc[GPBO_MAX_D];  // 1 / ls_k
};

// Reductions of a selection step with DPP row operations instead of ds_bpermute shuffles.  A step is issue-bound (sixteen
// waves share four SIMDs: every instruction of the step costs ~6 cycles x 4 waves), so the arg-max is taken in two cheap
// phases - the largest VALUE (two DPP moves + v_max_f64 per stage), then the lowest INDEX among the lanes that hold it
// (one DPP move + v_min_i32 per stage) - instead of one (value, index) compare-and-select chain per stage.
__device__ __forceinline__ bool fps_better(double v2, int i2, double v, int i) { return (v2 > v) || (v2 == v && i2 < i); }
template <int CTRL>
__device__ __forceinline__ double fps_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// every lane of a ROW of 16 gets the row's result: xor 1, xor 2 (quad_perm), then the half-row and row mirrors
__device__ __forceinline__ double fps_row_max(double v) {
    v = fmax(v, fps_dpp_f64<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmax(v, fps_dpp_f64<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmax(v, fps_dpp_f64<0x141>(v));   // row_half_mirror
    v = fmax(v, fps_dpp_f64<0x140>(v));   // row_mirror
    return v;
}
__device__ __forceinline__ int fps_row_min(int i) {
    i = min(i, __builtin_amdgcn_update_dpp(0, i, 0xB1, 0xf, 0xf, true));
    i = min(i, __builtin_amdgcn_update_dpp(0, i, 0x4E, 0xf, 0xf, true));
    i = min(i, __builtin_amdgcn_update_dpp(0, i, 0x141, 0xf, 0xf, true));
    i = min(i, __builtin_amdgcn_update_dpp(0, i, 0x140, 0xf, 0xf, true));
    return i;
}
// ... and every lane of the wave the wave's result: the four row results through scalar registers
__device__ __forceinline__ double fps_wave_max(double v) {
    v = fps_row_max(v);
    double r = v;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        r = fmax(r, __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * q),
                                     __builtin_amdgcn_readlane(__double2loint(v), 16 * q)));
    return fmax(r, __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0), __builtin_amdgcn_readlane(__double2loint(v), 0)));
}
__device__ __forceinline__ int fps_wave_min(int i) {
    i = fps_row_min(i);
    return min(min(__builtin_amdgcn_readlane(i, 0), __builtin_amdgcn_readlane(i, 16)),
               min(__builtin_amdgcn_readlane(i, 32), __builtin_amdgcn_readlane(i, 48)));
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

// ---- the selection with every observation in a REGISTER: one workgroup, or G workgroups that exchange one record per member ----
// Thread t of workgroup w keeps PTS observations (scaled coordinates, running minimum distance to the members) in registers:
// a selection step is D fmas per point, a wave reduction, one pass through LDS - and, with G > 1, ONE hand-off through
// memory: every workgroup publishes its best point (value, index AND coordinates) in its slot, every workgroup reads all G
// slots and picks the winner itself (all-to-all: no second hop to broadcast the result, no global load of the winner's
// coordinates).  Same arithmetic and tie rule as the one-launch-per-member form below (distance = fma chain over the
// coordinates in index order; largest value, lowest index), hence the same sequence.
//
// A step is bound by instruction ISSUE, not by latency: every wave runs the same ~65 instructions of reductions plus ~26 per
// point (d = 8), and the waves of a SIMD take turns - measured 1.15 / 1.4 / 1.9 us per member with 1,024 threads and 1 / 2 / 4
// points per thread.  Few waves with many points each are therefore the faster shape: 256 threads (one wave per SIMD) with up
// to 8 points per thread, 512 threads when one workgroup has to hold up to 4,096 points (TH, PTS below).
//
// The hand-off needs no fence: a record is 3 + 2 D WORDS of 64 bits, each a relaxed device-scope atomic that carries 32
// bits of payload AND the step's stamp (step + 1) - (value lo | stamp), (value hi | stamp), (index | stamp), then the
// halves of the coordinates.  A reader accepts a record when EVERY word shows the stamp it waits for: each word validates
// itself, so no ordering between words is assumed (no reliance on cache policy, no racy plain accesses),
// and no release fence - which on this chip writes the L2's dirty lines back, ~0.5 us per step - is paid.  The record is
// written by the winner's wave (lane w stores word w: one coalesced store) and read by wave 0 of every workgroup (lane w
// loads word w of all G slots: G coalesced loads per poll), so a hand-off is one store and one load on the wire.
// The workgroups that exchange are blockIdx.x = 0, 8, 16, ... of a grid of 8 G: under the round-robin dispatch they sit on
// ONE XCD (speed only; any placement is correct).  Two slot sets alternate by step parity: a workgroup
// can only be one step ahead of the slowest (it needs everybody's step-s record to start step s + 1), so the set of
// step s + 2 is never written while somebody still reads step s.  Every wait is BOUNDED: a workgroup that does not see a
// stamp within FPS_MAX_POLLS polls raises stt->error and leaves; so do the others; the host-side fall-back (identity
// order - the arrival order, still an exact route) is applied by fps_check_kernel.  Nothing can hang.
struct FpsSlot {
    unsigned long long w[64];   // 3 + 2 d <= 35 words used; 512 bytes: slots of different workgroups share no cache line
};
static_assert(sizeof(FpsSlot) == 512, "FpsSlot");
constexpr int FPS_MAXW = 16;            // cooperating workgroups at most
constexpr int FPS_MAX_POLLS = 1 << 21;  // ~1 s: a co-operating workgroup that was never scheduled

struct FpsState {
    int64_t member;        // newest member (one-launch-per-member form: read by the next launch)
    unsigned int ticket;   // workgroups of the current launch that have finished
    unsigned int error;    // cooperative form: a bounded wait ran out
    double centre[GPBO_MAX_D];   // scaled coordinates the next sweep measures distances to (first: the centroid)
};

template <int TH, int PTS, int D, bool COOP>
__global__ __launch_bounds__(TH) void fps_coop_kernel(const double *__restrict__ X, int64_t N, FpsLs ls, int64_t J, int G,
                                                      FpsState *__restrict__ stt, FpsSlot *__restrict__ slots,
                                                      double *__restrict__ mind, int64_t *__restrict__ perm,
                                                      int mute_wg /* tests: this workgroup never publishes (-1: none) */) {
    __shared__ double s_val[TH / 64];
    __shared__ int64_t s_idx[TH / 64];
    __shared__ double s_c[D];
    __shared__ int64_t s_member;
    __shared__ int s_dead;
    if (COOP && (blockIdx.x & 7)) return;   // (see above: the workers are every eighth workgroup)
    const int wg = COOP ? (int)(blockIdx.x >> 3) : 0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const double ninf = -std::numeric_limits<double>::infinity();
    const int64_t base = (int64_t)wg * TH * PTS;
    double xr[PTS][D], md[PTS];
#pragma unroll
    for (int q = 0; q < PTS; ++q) {
        const int64_t i = base + (int64_t)q * TH + tid;
#pragma unroll
        for (int k = 0; k < D; ++k) xr[q][k] = (i < N) ? X[i * D + k] * ls.isc[k] : 0.0;
        md[q] = (i < N) ? std::numeric_limits<double>::infinity() : ninf;   // rows beyond N are never chosen
    }
    if (tid < D) s_c[tid] = stt->centre[tid];
    if (tid == 0) { s_member = -1; s_dead = 0; }
    __syncthreads();
    for (int64_t s = 0; s <= J; ++s) {   // sweep s folds member s - 1 in (s = 0: distances to the centroid) and picks member s
        const int64_t member = s_member;
        double bv = ninf;
        int bi32 = 0x7fffffff;   // (N <= TH x PTS x 16 <= 65,536 here: the index fits 32 bits)
        int bq = 0;
#pragma unroll
        for (int q = 0; q < PTS; ++q) {
            const int64_t i = base + (int64_t)q * TH + tid;
            double dist = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double df = xr[q][k] - s_c[k];
                dist = fma(df, df, dist);
            }
            // a NaN coordinate (in this row, or - first sweep - anywhere: the centroid) gives a NaN distance, which no
            // comparison selects: such a row counts as -1, below every real distance and above a member's -inf, so a sweep
            // always has a winner with a valid index (the factorisation then reports the NaN; the order must not fault)
            if (!(dist == dist)) dist = -1.0;
            double m = dist;
            if (member >= 0) {
                m = fmin(md[q], dist);
                if (i == member) m = ninf;  // a member is never chosen again (its duplicates: distance 0, chosen last)
                md[q] = m;
            }
            if (i < N && fps_better(m, (int)i, bv, bi32)) { bv = m; bi32 = (int)i; bq = q; }
        }
        if (s == J) break;                  // the last sweep only marks member J - 1
        const int mine_i = bi32;            // this thread's own best (to recognise itself as the owner of the winner)
        // (comparisons never select a NaN distance and fmax drops it: same candidates either way)
        const double wmax = fps_wave_max(bv);
        if (lane == 0) s_val[w] = wmax;
        __syncthreads();
        const double gmax = fps_row_max(s_val[lane & (TH / 64 - 1)]);   // the TH / 64 wave results, repeated along every row of 16 lanes
        const int cand = (bv == gmax) ? bi32 : 0x7fffffff;              // lowest index among the points that attain it
        const int wmin = fps_wave_min(cand);
        if (lane == 0) s_idx[w] = wmin;
        __syncthreads();
        bi32 = fps_row_min((int)s_idx[lane & (TH / 64 - 1)]);
        bv = gmax;
        const int64_t bi = bi32 == 0x7fffffff ? std::numeric_limits<int64_t>::max() : (int64_t)bi32;
        // every thread now holds the workgroup's best (bv, bi); its owner has the coordinates in registers
        const bool have = bi != std::numeric_limits<int64_t>::max();
        const bool owner = have ? (mine_i == bi32) : (tid == 0);
        if (!COOP) {
            if (owner) {
#pragma unroll
                for (int q = 0; q < PTS; ++q)
                    if (bq == q) {
#pragma unroll
                        for (int k = 0; k < D; ++k) s_c[k] = xr[q][k];
                    }
                s_member = bi;
                perm[s] = bi;
            }
        } else {
            FpsSlot *set = slots + (size_t)(s & 1) * FPS_MAXW;
            constexpr int NWORD = 3 + 2 * D;
            const unsigned long long stamp = (unsigned long long)(unsigned)(s + 1) << 32;
            // the winner's wave publishes the record: the owner's registers reach the other lanes through scalar registers
            const int oloc = have ? (int)(bi - base) : 0;         // (a workgroup without a valid point: thread 0, value -inf)
            const int ow = (oloc % TH) >> 6, olane = oloc & 63, oq = oloc / TH;   // all uniform
            if (w == ow && wg != mute_wg) {
                unsigned data = 0;
                if (lane == 0) data = (unsigned)__double2loint(bv);
                if (lane == 1) data = (unsigned)__double2hiint(bv);
                if (lane == 2) data = (unsigned)bi32;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double ck = xr[0][k];
#pragma unroll
                    for (int q = 1; q < PTS; ++q) ck = (oq == q) ? xr[q][k] : ck;
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane(__double2loint(ck), olane);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane(__double2hiint(ck), olane);
                    if (lane == 3 + 2 * k) data = lo;
                    if (lane == 4 + 2 * k) data = hi;
                }
                if (lane < NWORD)
                    __hip_atomic_store(&set[wg].w[lane], stamp | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (w == 0) {
                // lane w polls word w of every slot; a poll = G loads in flight; done when every word carries this step's stamp
                const int wl = lane < NWORD ? lane : 0;
                unsigned r[FPS_MAXW];   // payload halves; the stamps are checked as the words arrive
                int polls = 0;
                bool timed_out = false;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int g = 0; g < FPS_MAXW; ++g) {
                        const unsigned long long v = (g < G) ? __hip_atomic_load(&set[g].w[wl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : stamp;
                        r[g] = (unsigned)v;
                        ok = ok && ((v >> 32) == (stamp >> 32));
                    }
                    if (__all(ok)) break;
                    if (++polls > FPS_MAX_POLLS) { timed_out = true; break; }
                    __builtin_amdgcn_s_sleep(1);   // (polling without it: no difference, 2.97 against 2.99 us per member)
                }
                // winner over the slots (uniform arithmetic on values read from lanes 0 .. 2), then its coordinates
                double gv = ninf;
                int gi = 0x7fffffff, gw = 0;
#pragma unroll
                for (int g = 0; g < FPS_MAXW; ++g) {
                    if (g < G) {
                        const int dlo = __builtin_amdgcn_readlane((int)r[g], 0);
                        const int dhi = __builtin_amdgcn_readlane((int)r[g], 1);
                        const int oi = __builtin_amdgcn_readlane((int)r[g], 2);
                        const double ov = __hiloint2double(dhi, dlo);
                        if (fps_better(ov, oi, gv, gi)) { gv = ov; gi = oi; gw = g; }
                    }
                }
                unsigned sel = r[0];
#pragma unroll
                for (int g = 1; g < FPS_MAXW; ++g) sel = (gw == g) ? r[g] : sel;
                if (lane >= 3 && lane < NWORD) reinterpret_cast<unsigned *>(s_c)[lane - 3] = sel;   // lo / hi halves in place
                if (lane == 0) {
                    s_member = gi == 0x7fffffff ? std::numeric_limits<int64_t>::max() : (int64_t)gi;
                    if (wg == 0) perm[s] = (int64_t)gi;
                    if (timed_out) {
                        s_dead = 1;
                        __hip_atomic_store(&stt->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        __syncthreads();
        if (s_dead) return;   // a bounded wait ran out: every workgroup leaves (fps_check_kernel repairs perm)
    }
#pragma unroll
    for (int q = 0; q < PTS; ++q) {   // the extension to the full order reads the membership from `mind`
        const int64_t i = base + (int64_t)q * TH + tid;
        if (i < N) mind[i] = md[q];
    }
}

// after the cooperative form: a bounded wait ran out (never seen) -> the identity order (arrival order: exact, prunes less)
__global__ __launch_bounds__(FT) void fps_check_kernel(const FpsState *__restrict__ stt, int64_t N, int64_t J,
                                                       double *__restrict__ mind, int64_t *__restrict__ perm) {
    if (!stt->error) return;
    const double ninf = -std::numeric_limits<double>::infinity();
    for (int64_t i = (int64_t)blockIdx.x * FT + threadIdx.x; i < N; i += (int64_t)gridDim.x * FT) {
        if (i < J) perm[i] = i;
        mind[i] = (i < J) ? ninf : 0.0;   // members 0 .. J-1; fps_extend_kernel appends the others in index order
    }
}

// ---- the same selection as ONE LAUNCH PER STEP, for sizes whose points do not fit one workgroup's registers --------------
// (8 points x 8 coordinates per thread of the 1024 spill: 12 us per step, 6.9 ms of sampling at N = 8192, d = 8; the
// single-workgroup global-memory form is slower still.)  Every workgroup folds the newest member into the running minimum
// of its own points and writes its best (value, index); the workgroup that finishes LAST - an atomic ticket, no waiting -
// reduces the partials in workgroup order and publishes the next member.  Same arithmetic, same tie rule (lowest index),
// hence the same sequence as fps_kernel; the cost is a kernel boundary per member (~2.5 us).
constexpr int FG = 256;

// centroid of the scaled observations, every coordinate in one pass: per-thread partial sums (rows tid, tid + 1024, ...), a
// butterfly over the lanes of each wave, then thread k adds the FT / 64 wave sums of coordinate k in wave order - a fixed
// order of additions, so the same centroid (and the same first member) every time.  (One coordinate after the other with a
// serial sum of the 1,024 partials cost 8 us per coordinate: 130 us of a 1.9-ms order at d = 16.)
__global__ __launch_bounds__(FT) void fps_centroid_kernel(const double *__restrict__ X, int64_t N, int d, FpsLs ls,
                                                          FpsState *__restrict__ stt) {
    __shared__ double s_w[FT / 64][GPBO_MAX_D];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double acc[GPBO_MAX_D];
#pragma unroll
    for (int k = 0; k < GPBO_MAX_D; ++k) acc[k] = 0.0;
    for (int64_t i = tid; i < N; i += FT) {
#pragma unroll
        for (int k = 0; k < GPBO_MAX_D; ++k)
            if (k < d) acc[k] += X[i * d + k] * ls.isc[k];
    }
#pragma unroll
    for (int k = 0; k < GPBO_MAX_D; ++k) {
        if (k < d) {   // (uniform)
            double v = acc[k];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off);
            if (lane == 0) s_w[w][k] = v;
        }
    }
    __syncthreads();
    if (tid < d) {
        double t = 0.0;
        for (int q = 0; q < FT / 64; ++q) t += s_w[q][tid];
        stt->centre[tid] = t / (double)N;
    }
    if (tid == 0) { stt->member = -1; stt->ticket = 0; stt->error = 0; }
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
        if (!(dist == dist)) dist = -1.0;   // NaN coordinates: see fps_coop_kernel
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
        // The partials are published by a RELEASE on the ticket and read behind an ACQUIRE fence by the workgroup that takes
        // the last one (a proper release / acquire edge instead of relying on the cache policy of sc1 stores).
        // The release writes this XCD's dirty L2 lines back (the running minima just stored): ~1 us per launch, accepted -
        // since round 4 this form only serves N > 32,768, everything smaller runs in fps_coop_kernel.
        __hip_atomic_store(pval + blockIdx.x, bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pidx + blockIdx.x, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (__hip_atomic_fetch_add(&stt->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (s_last) {  // every workgroup's partial has been released: reduce them (the order does not matter: largest value,
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
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

// (an index outside [0, n) cannot come out of the selection; if it ever did, the row is NaN - the factorisation then fails
//  loudly - instead of an out-of-bounds read)
__global__ void gather_obs_kernel(const double *__restrict__ X, int d, const int64_t *__restrict__ perm, int64_t n,
                                  double *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * d) return;
    const int64_t r = perm[e / d];
    out[e] = (r >= 0 && r < n) ? X[r * d + (e % d)] : std::numeric_limits<double>::quiet_NaN();
}

struct OrderLayout {
    int64_t mind_off, fps_off, slot_off, total;
};
constexpr int FPS_MAXG = 256;  // workgroups of a selection step at most

OrderLayout order_layout(int64_t N) {
    OrderLayout L;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    L.mind_off = take((int64_t)sizeof(double) * N);
    L.fps_off = take((int64_t)sizeof(FpsState) + (int64_t)FPS_MAXG * (sizeof(double) + sizeof(int64_t)) + 256);
    L.slot_off = take((int64_t)sizeof(FpsSlot) * 2 * FPS_MAXW);
    L.total = off;
    return L;
}

__global__ void gather_y_kernel(const double *__restrict__ y, const int64_t *__restrict__ perm, int64_t n, double *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) {
        const int64_t r = perm[e];
        out[e] = (r >= 0 && r < n) ? y[r] : std::numeric_limits<double>::quiet_NaN();
    }
}

}  // namespace

extern "C" int64_t gpbo_fps_order_workspace_bytes(int64_t N) {
    if (N < 1) return GPBO_ERR_ARG;
    return order_layout(N).total;
}

// 1 into *fell_back (device int32) when the last gpbo_fps_order_f64 on this workspace gave up its co-operative selection (a
// bounded wait ran out: a workgroup was never scheduled) and installed the arrival order instead - still an exact route, but
// a caller whose peers must hold the SAME factorisation (candidate shards of one step) has to know.  Enqueued on `stream`.
namespace {
__global__ void fps_status_kernel(const FpsState *__restrict__ stt, int32_t *__restrict__ out) { *out = stt->error ? 1 : 0; }
}  // namespace

extern "C" int gpbo_fps_order_status(const void *work, int64_t N, int32_t *fell_back, void *stream) {
    if (!work || !fell_back || N < 1 || ((uintptr_t)work & 255)) return GPBO_ERR_ARG;
    const OrderLayout L = order_layout(N);
    const FpsState *stt = reinterpret_cast<const FpsState *>(reinterpret_cast<const char *>(work) + L.fps_off);
    hipLaunchKernelGGL(fps_status_kernel, dim3(1), dim3(1), 0, gpbo_stream(stream), stt, fell_back);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
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
    FpsState *stt = reinterpret_cast<FpsState *>(w + L.fps_off);
    // Shape of the selection (see fps_coop_kernel): PTS points per thread, as many as 128 registers of coordinates hold
    // (8 up to d = 8, 4 above); one workgroup of 256 threads while it can hold the points, of 512 up to 4,096 (2,048)
    // points.  Beyond, workgroups of 512 threads share the points, EIGHT of them while 8 x 512 x pmax points suffice
    // (measured at N = 8192, d = 8, us per member: 512 x 2 x 8 workgroups 2.96, 256 x 4 x 8 3.13, 1024 x 1 x 8 3.2,
    // 256 x 2 x 16 3.43, 256 x 8 x 4 3.48: the hand-off grows with the number of slots a reader polls, the local part
    // with the points per SIMD), sixteen up to 16 x 512 x pmax = 65,536 (32,768) points.  Beyond those 16 workgroups: one
    // launch per member (fps_step_kernel).
    const int pmax = d <= 8 ? 8 : 4;
    int th = 256, pts = 1;
    while (pts < pmax && (int64_t)th * pts < N) pts *= 2;
    if ((int64_t)th * pts < N) th = 512;
    bool coop = (int64_t)th * pts < N;
    if (coop) {
        th = 512;
        pts = 1;
        while (pts < pmax && (int64_t)th * pts * 8 < N) pts *= 2;
    }
    int mute_wg = -1;
#ifdef GPBO_DIAGNOSTICS
    // Diagnostics builds only (never the shipped library; tests load one through GPBO_LIB):
    // GPBO_FPS_MUTE=k: co-operating workgroup k never publishes its record, as if it had never been scheduled - the others'
    // bounded waits run out (~1 s), every workgroup leaves, fps_check_kernel installs the identity order (the arrival order:
    // an exact route, it only prunes less) and gpbo_fps_order_status reports the fall-back.
    // GPBO_FPS_SHAPE="threads,points" overrides the shape (A/B runs; ignored when the points do not fit).
    static const int mute_env = getenv("GPBO_FPS_MUTE") ? atoi(getenv("GPBO_FPS_MUTE")) : -1;
    static const char *shape_env = getenv("GPBO_FPS_SHAPE");
    mute_wg = mute_env;
    if (shape_env) {
        int eth = 0, epts = 0;
        if (sscanf(shape_env, "%d,%d", &eth, &epts) == 2 && (eth == 256 || eth == 512) &&
            (epts == 1 || epts == 2 || epts == 4 || epts == 8) && epts <= pmax && (int64_t)eth * epts * FPS_MAXW >= N) {
            th = eth;
            pts = epts;
            coop = (int64_t)th * pts < N;
        }
    }
#endif
    const int64_t G = (N + (int64_t)th * pts - 1) / ((int64_t)th * pts);
    bool launched = false;
    if (G <= FPS_MAXW) {
        FpsSlot *slots = reinterpret_cast<FpsSlot *>(w + L.slot_off);
        if (coop && hipMemsetAsync(slots, 0, sizeof(FpsSlot) * 2 * FPS_MAXW, st) != hipSuccess) return GPBO_ERR_LAUNCH;
        hipLaunchKernelGGL(fps_centroid_kernel, dim3(1), dim3(FT), 0, st, X, N, (int)d, ls, stt);
        const unsigned grid = coop ? (unsigned)(8 * G) : 1u;
#define GPBO_FPS(T, P, DD, CO) hipLaunchKernelGGL((fps_coop_kernel<T, P, DD, CO>), dim3(grid), dim3(T), 0, st, X, N, ls, J, (int)G, stt, slots, mind, perm_out, mute_wg)
#define GPBO_FPS_P(T, DD, CO)                                                      \
    do {                                                                            \
        if (pts == 1) GPBO_FPS(T, 1, DD, CO);                                       \
        else if (pts == 2) GPBO_FPS(T, 2, DD, CO);                                  \
        else if (pts == 4 || DD > 8) GPBO_FPS(T, 4, DD, CO);                        \
        else GPBO_FPS(T, (DD <= 8 ? 8 : 4), DD, CO);                                \
    } while (0)
#define GPBO_FPS_D(DD)                                                              \
    if (!launched && d == DD) {                                                     \
        if (th == 256) { if (coop) GPBO_FPS_P(256, DD, true); else GPBO_FPS_P(256, DD, false); }  \
        else { if (coop) GPBO_FPS_P(512, DD, true); else GPBO_FPS_P(512, DD, false); }            \
        launched = true;                                                            \
    }
        GPBO_FPS_D(1) GPBO_FPS_D(2) GPBO_FPS_D(3) GPBO_FPS_D(4) GPBO_FPS_D(5) GPBO_FPS_D(6) GPBO_FPS_D(7) GPBO_FPS_D(8)
        GPBO_FPS_D(9) GPBO_FPS_D(10) GPBO_FPS_D(11) GPBO_FPS_D(12) GPBO_FPS_D(13) GPBO_FPS_D(14) GPBO_FPS_D(15) GPBO_FPS_D(16)
#undef GPBO_FPS_D
#undef GPBO_FPS_P
#undef GPBO_FPS
        if (coop) hipLaunchKernelGGL(fps_check_kernel, dim3(8), dim3(FT), 0, st, stt, N, J, mind, perm_out);
        hipLaunchKernelGGL(fps_extend_kernel, dim3(1), dim3(FT), 0, st, N, J, N, mind, perm_out);
    }
    if (!launched) {
        // one launch per member (the state block: FpsState, then the workgroups' partial values and indices)
        double *pval = reinterpret_cast<double *>(w + L.fps_off + ((sizeof(FpsState) + 255) / 256) * 256);
        int64_t *pidx = reinterpret_cast<int64_t *>(pval + FPS_MAXG);
        int64_t G = (N + FG - 1) / FG;
        if (G > FPS_MAXG) G = FPS_MAXG;
        hipLaunchKernelGGL(fps_centroid_kernel, dim3(1), dim3(FT), 0, st, X, N, (int)d, ls, stt);
        for (int64_t j = 0; j <= J; ++j)   // launch j: folds member j - 1 in and picks member j (the last one only folds)
            hipLaunchKernelGGL(fps_step_kernel, dim3((unsigned)G), dim3(FG), 0, st, X, N, (int)d, ls, mind, stt, pval, pidx, perm_out, j, J);
        hipLaunchKernelGGL(fps_extend_kernel, dim3(1), dim3(FT), 0, st, N, J, N, mind, perm_out);
    }
    if (Xp_out) {
        const int64_t tot = N * d;
        hipLaunchKernelGGL(gather_obs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, X, (int)d, perm_out, N, Xp_out);
    }
    if (yp_out) hipLaunchKernelGGL(gather_y_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, y, perm_out, N, yp_out);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}
