// fp32 scoring path (BASELINE.json config 4: d=16, N=8192, fp32 candidates).
//
// Same mathematics as kernel_build.hip / sigma_acq.hip (reference: point_selector.py:81, :90-98, :204-207)
// with the N^2-per-candidate work - the triangular product - in fp32 on v_mfma_f32_16x16x4_f32.
// K(X*,X) entries and the mean are computed in fp64 exactly as in the fp64 path (kernel_build.hip,
// gpbo_kstar_mu_mixed: the mean is the fp64 path's mean bit for bit; 2N of the N^2 flops per candidate) and only the
// stored K*^T is rounded to fp32.  The factorisation stays in fp64 (cond(K) ~ 1e6 with the reference's 1.01e-4
// jitter: an fp32 Cholesky at N = 8192 meets non-positive pivots) and U is rounded to fp32 once per BO step by
// gpbo_prepare_f32.  This pass is a SCREEN: its variance carries fp32 error, so the arg-max is decided by
// rescore.hip, which re-scores every candidate that could still be the maximum through the fp64 kernels.
//
// (Column groups on one XCD, which give the fp64 and int8 kernels 6-9 %, were measured here too: nothing at N = 8192,
//  269.4 against 270.0 ms per step - the fp32 copy of U alone is 268 MB, more than the Infinity Cache holds.)
// Variance kernel geometry (differs from the fp64 one because elements are 4 bytes):
//   workgroup 512 threads, 256 candidates x 256 columns of V, 16-deep k tiles, 3-stage LDS ring fed by
//   global_load_lds_dwordx4 (one 1-KiB piece = one 256-float row); wave tile 64 x 128 = 4 x 8 MFMA tiles;
//   C/D map of the f32 MFMA: result register r of lane l is row 4*(l>>4) + r, column l&15.
#include "gpbo_internal.h"

#include <cstdlib>
#include <limits>
#include <type_traits>

namespace {

typedef float f4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int BM = 256, BN = 256, BK = 16;
constexpr int WR = 4, WQ = 2, NI = 8;  // 8 waves: 4 row groups of 64 candidates x 2 column groups; 8 column tiles each
constexpr int LDA = BM + 16, LDB = BN + 16;
constexpr int A_TILE = BK * LDA, B_TILE = BK * LDB, STAGE = A_TILE + B_TILE;  // floats
constexpr int KS_SLICE = 64;

struct LsArgs32 {
    double isc[GPBO_MAX_D];  // 1 / (ls_k sqrt 2)
};

__device__ __forceinline__ void glds16f(const float *g, float *l) {
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)l, 16, 0, 0);
}

__device__ __forceinline__ bool better(double v2, int64_t i2, double v, int64_t i) { return gpbo_better(v2, i2, v, i); }

// ---- U32 / alpha32: fp32 copies of the fp64 factors, re-padded to Np32 (identity / zeros on the padding)
__global__ __launch_bounds__(256) void prepare_f32_kernel(const double *__restrict__ U, const double *__restrict__ alpha,
                                                         int64_t Np, float *__restrict__ U32, float *__restrict__ alpha32,
                                                         int64_t Np32) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < Np32 * Np32) {
        const int64_t r = e / Np32, c = e - r * Np32;
        float v = (r == c) ? 1.0f : 0.0f;
        if (r < Np && c < Np) v = (float)U[r * Np + c];
        U32[e] = v;
    }
    if (alpha32 && e < Np32) alpha32[e] = (e < Np) ? (float)alpha[e] : 0.0f;
}

// ---- variance + acquisition + block arg-max, fp32 MFMA
__global__ __launch_bounds__(512) void sigma_acq_f32_kernel(
    const float *__restrict__ KsT, int64_t ldk, const float *__restrict__ U, int Np, const double *__restrict__ mu_part,
    int nsl, int64_t Mc, double prior_var, int acq_kind, double p0, double p1, int64_t idx_base,
    double *__restrict__ mu_out, double *__restrict__ sigma_out, double *__restrict__ acq_out,
    double *__restrict__ var_out /* signed prior_var - |v|^2 as the fp32 product leaves it (the screen's input) */,
    double *__restrict__ part_val, int64_t *__restrict__ part_idx, unsigned long long *__restrict__ nan_count) {
    __shared__ float smem[3 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid % WR, wq = wid / WR;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t cand0 = (int64_t)blockIdx.x * BM;
    const float *a_src = KsT + cand0;
    const float *b_src = U;
    const int lane4 = lane * 4;

    f4_t acc[4][NI];
    float ss[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) ss[i][r] = 0.f;
    }
    const int nJ = Np / BN;
    // one piece = one 256-float row (1 KiB): 16 A rows + 16 B rows per tile, wave w takes pieces w + 8r
    auto stage = [&](int jb, int kt, int buf) {
        float *As = smem + buf * STAGE;
        float *Bs = As + A_TILE;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = wid + 8 * r;
            if (u < 16) glds16f(a_src + (int64_t)(kt * BK + u) * ldk + lane4, As + u * LDA);
            else glds16f(b_src + (int64_t)(kt * BK + u - 16) * Np + jb * BN + lane4, Bs + (u - 16) * LDB);
        }
    };
    auto advance = [&](int &j, int &k) {
        if (++k == (j + 1) * (BN / BK)) { ++j; k = 0; }
    };
    int pj = 0, pk = 0, pbuf = 0;
    stage(pj, pk, pbuf);
    advance(pj, pk);
    pbuf = 1;
    if (pj < nJ) {
        stage(pj, pk, pbuf);
        advance(pj, pk);
        pbuf = 2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    float a0[4], b0[NI], a1[4], b1[NI];
    auto lds_frag = [&](float (&af)[4], float (&bf)[NI], int buf, int kk) {
        const float *As = smem + buf * STAGE;
        const float *Bs = As + A_TILE;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi] = As[(kk + l4) * LDA + wr * 64 + mi * 16 + l15];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bf[ni] = Bs[(kk + l4) * LDB + (WQ * ni + wq) * 16 + l15];
    };
    lds_frag(a0, b0, 0, 0);
    int cur = 0;
    // FULL tiles (left of the diagonal block) run a branch-free MFMA stream; see sigma_acq.hip
    auto tile_body = [&](auto full_tag, int jb, int kt) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int nxt = (cur == 2) ? 0 : cur + 1;
        int ni_min = (kt - jb * (BN / BK) - wq + WQ - 1) / WQ;  // column tile WQ ni + wq is needed iff >= kt'
        ni_min = ni_min < 0 ? 0 : ni_min;
        auto mfma_half = [&](const float (&af)[4], const float (&bf)[NI], int nlo) {
#pragma unroll
            for (int ni = nlo; ni < nlo + NI / 2; ++ni) {
                if (FULL || ni >= ni_min) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
                }
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a0, b0, 0);
        lds_frag(a1, b1, cur, 4);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a0, b0, NI / 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a1, b1, 0);
        lds_frag(a0, b0, cur, 8);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a1, b1, NI / 2);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own pieces of tile t+1 have landed
        __builtin_amdgcn_s_barrier();
        // column group 0 issues its DMA pieces of tile t+2 here, column group 1 at the end of the tile, so that the
        // two waves of a SIMD do not stop feeding the matrix pipe at the same moment
        const bool do_stage = pj < nJ;
        auto stage_adv = [&]() {
            stage(pj, pk, pbuf);
            advance(pj, pk);
            pbuf = (pbuf == 2) ? 0 : pbuf + 1;
        };
        if (do_stage && wq == 0) stage_adv();
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a0, b0, 0);
        lds_frag(a1, b1, cur, 12);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a0, b0, NI / 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a1, b1, 0);
        lds_frag(a0, b0, nxt, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a1, b1, NI / 2);
        __builtin_amdgcn_sched_barrier(0);
        if (do_stage && wq != 0) stage_adv();
        cur = nxt;
    };
    for (int jb = 0; jb < nJ; ++jb) {
        const int heavy = jb * (BN / BK);
        for (int kt = 0; kt < heavy; ++kt) tile_body(std::true_type{}, jb, kt);
        for (int kt = heavy; kt < heavy + BN / BK; ++kt) tile_body(std::false_type{}, jb, kt);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ss[mi][r] = fmaf(acc[mi][ni][r], acc[mi][ni][r], ss[mi][r]);
                acc[mi][ni] = f4_t{0.f, 0.f, 0.f, 0.f};
            }
    }

    __syncthreads();
    float *red = smem;  // [WQ][BM]
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = ss[mi][r];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if (l15 == 0) red[wq * BM + wr * 64 + mi * 16 + 4 * l4 + r] = v;  // f32 MFMA C/D row map
        }
    __syncthreads();
    double *s_val = reinterpret_cast<double *>(smem + WQ * BM);
    int64_t *s_idx = reinterpret_cast<int64_t *>(smem + WQ * BM + 16);
    if (tid < BM) {
        const int64_t c = cand0 + tid;
        const bool valid = c < Mc;
        const float ssq = red[tid] + red[BM + tid];
        double mu = 0.0;  // same fixed-order sum of the same fp64 partials as the fp64 path
        for (int s = 0; s < nsl; ++s) mu += mu_part[(int64_t)s * ldk + c];
        const double var = prior_var - (double)ssq;
        const double sigma = sqrt(fabs(var));
        const double acq = gpbo_acquisition(acq_kind, mu, sigma, p0, p1);
        if (valid) {
            if (mu_out) mu_out[c] = mu;
            if (sigma_out) sigma_out[c] = sigma;
            if (acq_out) acq_out[c] = acq;
            if (var_out) var_out[c] = var;
        }
        const bool is_nan = valid && (acq != acq);
        const unsigned long long nan_mask = __ballot(is_nan);
        if (lane == 0 && nan_mask) atomicAdd(nan_count, (unsigned long long)__popcll(nan_mask));
        double bv = (valid && !is_nan) ? acq : -std::numeric_limits<double>::infinity();
        int64_t bi = (valid && !is_nan) ? idx_base + c : std::numeric_limits<int64_t>::max();
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double ov = __shfl_xor(bv, off);
            const int64_t oi = __shfl_xor(bi, off);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_val[tid >> 6] = bv; s_idx[tid >> 6] = bi; }
    }
    __syncthreads();
    if (tid == 0) {
        double bv = s_val[0];
        int64_t bi = s_idx[0];
        for (int w = 1; w < BM / 64; ++w)
            if (better(s_val[w], s_idx[w], bv, bi)) { bv = s_val[w]; bi = s_idx[w]; }
        part_val[blockIdx.x] = bv;
        part_idx[blockIdx.x] = bi;
    }
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct Layout32 {
    int64_t kst_off, mup_off, xsc_off, pval_off, pidx_off, nan_off, total, nparts_cap;
};

Layout32 layout32(int64_t Np32, int64_t chunk, int64_t M) {
    Layout32 L;
    const int64_t nchunks = (M + chunk - 1) / chunk;
    L.nparts_cap = nchunks * (chunk / BM);
    int64_t off = 0;
    L.kst_off = off; off += align_up((int64_t)sizeof(float) * Np32 * chunk, 256);
    L.mup_off = off; off += align_up((int64_t)sizeof(double) * (Np32 / KS_SLICE) * chunk, 256);
    L.xsc_off = off; off += align_up((int64_t)sizeof(double) * Np32 * GPBO_MAX_D, 256);
    L.pval_off = off; off += align_up((int64_t)sizeof(double) * L.nparts_cap, 256);
    L.pidx_off = off; off += align_up((int64_t)sizeof(int64_t) * L.nparts_cap, 256);
    L.nan_off = off; off += 256;
    L.total = off;
    return L;
}

}  // namespace

extern "C" int64_t gpbo_padded_n_f32(int64_t N) {
    if (N < 1) N = 1;
    return (N + BN - 1) / BN * BN;
}

extern "C" int gpbo_prepare_f32(const double *U, const double *alpha, int64_t Np, float *U32, float *alpha32,
                                int64_t Np32, void *stream) {
    if (!U || !U32 || (alpha32 && !alpha) || Np < 1 || Np32 < Np || Np32 % BN) return GPBO_ERR_ARG;
    const int64_t tot = Np32 * Np32;
    hipLaunchKernelGGL(prepare_f32_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, gpbo_stream(stream), U, alpha,
                       Np, U32, alpha32, Np32);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

extern "C" int64_t gpbo_posterior_workspace_bytes_f32(int64_t Np32, int64_t chunk, int64_t M) {
    if (Np32 < BN || Np32 % BN || chunk < 1024 || chunk % 1024 || chunk > GPBO_CHUNK_MAX || M < 1) return GPBO_ERR_ARG;
    return layout32(Np32, chunk, M).total;
}

extern "C" int gpbo_posterior_acq_f32(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np32, int32_t d,
                                      const double *ls_host, const float *U32, const double *alpha, double prior_var,
                                      int32_t acq_kind, double p0, double p1, double diag_add, int64_t idx_offset,
                                      int64_t chunk, double *mu_out, double *sigma_out, double *acq_out, double *var_out,
                                      gpbo_result *result, void *work, int64_t work_bytes, gpbo_profile *prof,
                                      void *stream) {
    if (!Xs || !X || !U32 || !alpha || !result || !work) return GPBO_ERR_ARG;
    if (M < 1 || N < 1 || Np32 != gpbo_padded_n_f32(N)) return GPBO_ERR_ARG;
    if (chunk < 1024 || chunk % 1024 || chunk > GPBO_CHUNK_MAX) return GPBO_ERR_ARG;
    if (acq_kind != GPBO_ACQ_LCB && acq_kind != GPBO_ACQ_EI) return GPBO_ERR_ARG;
    if (((uintptr_t)work & 255) || ((uintptr_t)U32 & 15)) return GPBO_ERR_ARG;
    const Layout32 L = layout32(Np32, chunk, M);
    if (work_bytes < L.total) return GPBO_ERR_WORKSPACE;
    hipStream_t st = gpbo_stream(stream);
    char *w = reinterpret_cast<char *>(work);
    float *KsT = reinterpret_cast<float *>(w + L.kst_off);
    double *mu_part = reinterpret_cast<double *>(w + L.mup_off);
    double *Xsc = reinterpret_cast<double *>(w + L.xsc_off);
    double *part_val = reinterpret_cast<double *>(w + L.pval_off);
    int64_t *part_idx = reinterpret_cast<int64_t *>(w + L.pidx_off);
    unsigned long long *nan_count = reinterpret_cast<unsigned long long *>(w + L.nan_off);
    // observations / (ls sqrt 2) in fp64, once per call; the same launch clears the NaN counter
    int rc = gpbo_scale_points_launch(X, N, Np32, d, ls_host, Xsc, nan_count, stream);
    if (rc != GPBO_OK) return rc;
    int64_t nparts = 0;
    bool prev_recorded = false;
    for (int64_t s = 0; s < M; s += chunk) {
        const int64_t Mc = (M - s < chunk) ? (M - s) : chunk;
        // K(X*,X) launches are timed like the fp64 path's: the launches of this call form one chain on the stream
        const bool krec = prof && prof->count < prof->capacity;
        if (krec) {
            const bool chained = s > 0 && prof->count > 0 && prev_recorded;
            prof->kmode[prof->count] = chained ? 2 : 1;
            if (!chained && hipEventRecord(reinterpret_cast<hipEvent_t>(prof->kbegin[prof->count]), st) != hipSuccess)
                return GPBO_ERR_LAUNCH;
        }
        rc = gpbo_kstar_mu_mixed(Xs + s * d, Mc, Xsc, N, Np32, d, ls_host, alpha, diag_add, idx_offset + s, KsT, chunk,
                                 mu_part, stream);
        if (rc != GPBO_OK) return rc;
        const int64_t nblk = (Mc + BM - 1) / BM;
        const bool rec = krec;
        if (rec && hipEventRecord(reinterpret_cast<hipEvent_t>(prof->begin[prof->count]), st) != hipSuccess)
            return GPBO_ERR_LAUNCH;
        hipLaunchKernelGGL(sigma_acq_f32_kernel, dim3((unsigned)nblk), dim3(512), 0, st, KsT, chunk, U32, (int)Np32, mu_part,
                           (int)(Np32 / KS_SLICE), Mc, prior_var, (int)acq_kind, p0, p1, idx_offset + s,
                           mu_out ? mu_out + s : nullptr, sigma_out ? sigma_out + s : nullptr,
                           acq_out ? acq_out + s : nullptr, var_out ? var_out + s : nullptr, part_val + nparts,
                           part_idx + nparts, nan_count);
        if (rec) {
            if (hipEventRecord(reinterpret_cast<hipEvent_t>(prof->end[prof->count]), st) != hipSuccess)
                return GPBO_ERR_LAUNCH;
            prof->cands[prof->count] = Mc;
            ++prof->count;
        }
        prev_recorded = rec;
        GPBO_CHECK_LAUNCH();
        nparts += nblk;
    }
    return gpbo_launch_argmax_finish(part_val, part_idx, nparts, nan_count, result, st);
}
