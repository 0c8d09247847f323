// Fused Cholesky + inverse factor of cov_meas on the gfx950 fp64 matrix cores.
//
// The reference inverts cov_meas with np.linalg.inv (/root/reference/point_selector.py:89) and uses the inverse in the
// posterior mean and covariance (:90-91).  The variance kernel (sigma_acq.hip) wants U = L^-T.  Round 2 built it in two
// passes (blocked Cholesky, then a block-recursive triangular inverse whose last level cannot start before the
// Cholesky has ended) out of ~200 dependent launches per factorisation.  This file replaces both passes by ONE sweep of
// row operations over the stacked matrix S = [A | W] (cholinv_plan.h): on exit W = L^-1 = U^T.
//
// One kernel, three kinds of workgroup (512 threads), mixed inside a launch (cholinv_plan.h decides):
//   PAIR       the 128 x 128 diagonal block of a pair of block rows: Cholesky AND the inverse of its factor (two
//              64 x 64 eliminations in registers + LDS + MFMA with the Schur step between them), computed redundantly
//              by every workgroup of the launch, then rows <- inv(L) * rows for the workgroup's own 64 columns, in
//              registers on the matrix cores.  The critical path of the factorisation.
//   UPD_SMALL  64 x 64 tile of out -= A^T B, K <= 256: MFMA fragments straight from global memory (k-major operands:
//              16 lanes read 128 contiguous bytes), no LDS, no barrier - the lowest latency for the updates that sit on
//              the critical path (the next pair's rows).
//   UPD_BIG    128 x 128 tile, 16-deep k tiles through a three-stage LDS ring filled by global_load_lds (the loop of
//              sigma_acq_kernel): the bulk of the flops, run as filler workgroups beside the panels.
// Dependencies are launch order on one stream; inside a launch no workgroup reads what another one writes
// (tests/test_cholinv_plan_cpu.py checks that on the plan itself).
#include "gpbo_internal.h"

#include "cholinv_plan.h"

#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>

namespace {

constexpr int NB = 64;
constexpr int PB = 16;               // inner block of the diagonal elimination
constexpr int LDM = NB + 2;          // row stride of the elimination matrix in LDS (doubles)
constexpr int LDT = 128 + 16;        // row stride of a 128-wide k-major operand tile in LDS: lanes l and l+16 of a
                                     // ds_read_b64 group (k and k+1) land in disjoint banks
constexpr int BKT = 16;              // k depth of a stage
constexpr int STAGE_D = BKT * (256 + 16 + LDT);        // doubles per stage of the widest ring: A tile [16][272] + B tile [16][144]
constexpr int PANEL_D = 4 * NB * LDM;                  // PAIR: four [64][LDM] blocks
constexpr int SMEM_D = (3 * STAGE_D > PANEL_D) ? 3 * STAGE_D : PANEL_D;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void glds16(const double *g, double *l) {
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)l, 16, 0, 0);
}

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

// A launch whose tiles are ONE regular rank-K update in 64 x w tiles (the NEAR launch of every pair) carries its shape in the
// kernel arguments: the workgroup forms its tile from blockIdx instead of reading the tile table - a dependent scalar load
// from memory (~1 us) before the first fragment can be requested, on the critical path of every pair.
struct CiNear {
    int32_t on, k0, K, row0, wlim, w, n0, r1;  // tile i: row block i >= n0, column row0 + 64 (i >= n0) + w (i - n0 (i >= n0))
};

struct CiArgs {
    double *S;
    int64_t ld;
    int32_t Np;
    int32_t *info;
    unsigned long long *stamps;  // GPBO_DIAGNOSTICS builds: cycle stamps of the first PAIR workgroup of a launch
    const CiTile *tab;           // the plan's tile table (device)
    CiLaunch l;
    CiNear near;
};

// ---------------------------------------------------------------------------------------------------------------
// UPD_BIG: out[BM x 128] -= sum_k S[k][row0 + m] * S[k][col0 + n], k in [k0, k0 + K);  BM = 64 MI = 128 or 256.
// 8 waves as 4 (rows) x 2 (columns): wave tile (BM/4) x 64 = MI x 4 MFMA tiles; the two waves of a SIMD (w, w + 4) take
// the even / odd 16-column tiles.  Per 16-deep k tile: 16 (BM + 128) doubles by LDS-DMA (1 KiB per wave instruction),
// 8 MI MFMAs per wave and k step; one barrier in the middle of the tile (see sigma_acq_kernel for why there).
// BM = 256 is the tile of sigma_acq_kernel (6 B/clk/CU of operands); BM = 128 needs 8 B/clk/CU and has half the
// MFMAs between two barriers: it serves the updates that have few tiles.
// ---------------------------------------------------------------------------------------------------------------
template <int MI>
__device__ __forceinline__ void upd_big(double *__restrict__ S, int64_t ld, int Np, const CiTile &u, double *smem,
                                        unsigned long long *stamps) {
    const bool nt = Np >= 6144;
#ifdef GPBO_DIAGNOSTICS
    int nst = 0;
#define CI_TSTAMP() do { if (stamps && threadIdx.x == 0) stamps[nst++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CI_TSTAMP() do { } while (0)
#endif
    CI_TSTAMP();
    constexpr int BM = 64 * MI;
    constexpr int LDA = BM + 16;                  // A tile row stride (doubles)
    constexpr int STG = BKT * (LDA + LDT);        // doubles per stage
    constexpr int PA = BM / 128;                  // 1 KiB pieces per A row
    constexpr int NPA = BKT * PA / 8, NPB = BKT / 8;  // pieces per wave per k tile
    static_assert(3 * STG <= SMEM_D, "LDS ring");
    const int row0 = u.row0, col0 = u.col0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid & 3, wq = wid >> 2;
    const int l15 = lane & 15, l4 = lane >> 4;

    const double *pa = S + (int64_t)u.k0 * ld + row0;
    const double *pb = S + (int64_t)u.k0 * ld + col0;
    unsigned voffA[NPA], voffB[NPB];
    int ldsA[NPA], ldsB[NPB];
#pragma unroll
    for (int r = 0; r < NPA; ++r) {
        const int v = wid + 8 * r, row = v / PA, piece = v % PA;
        voffA[r] = (unsigned)(row * (unsigned)ld + piece * 128 + lane * 2);
        ldsA[r] = row * LDA + piece * 128;
    }
#pragma unroll
    for (int r = 0; r < NPB; ++r) {
        voffB[r] = (unsigned)((wid + 8 * r) * (unsigned)ld + lane * 2);
        ldsB[r] = BKT * LDA + (wid + 8 * r) * LDT;
    }
    const int nk = u.K / BKT;
    int pk = 0, pbuf = 0;
    auto stage_next = [&]() {
        double *St = smem + pbuf * STG;
#pragma unroll
        for (int r = 0; r < NPA; ++r) glds16(pa + voffA[r], St + ldsA[r]);
#pragma unroll
        for (int r = 0; r < NPB; ++r) glds16(pb + voffB[r], St + ldsB[r]);
        pbuf = (pbuf == 2) ? 0 : pbuf + 1;
        ++pk;
        pa += (int64_t)BKT * ld;
        pb += (int64_t)BKT * ld;
    };

    d4_t acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};

    stage_next();
    if (pk < nk) stage_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    CI_TSTAMP();

    double a0[MI], b0[4], a1[MI], b1[4];
    auto lds_frag = [&](double (&af)[MI], double (&bf)[4], int buf, int kk) {
        const double *As = smem + buf * STG;
        const double *Bs = As + BKT * LDA;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[mi] = As[(kk + l4) * LDA + wr * (BM / 4) + mi * 16 + l15];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bf[ni] = Bs[(kk + l4) * LDT + (2 * ni + wq) * 16 + l15];
    };
    auto mfma_half = [&](const double (&af)[MI], const double (&bf)[4], int nlo) {
#pragma unroll
        for (int ni = nlo; ni < nlo + 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = mfma_f64_16x16x4(af[mi], bf[ni], acc[mi][ni]);
    };
    // The C tile is streamed INTO the accumulators while the k loop runs: row batch mi (its 16 live values per lane) is
    // loaded during k tile mi and subtracted from the accumulators during k tile mi + 1, so acc = sum - C and the
    // epilogue is nothing but the (posted) stores of -acc.  Read after the loop instead, the C tiles of all workgroups
    // of a launch cross the fabric together and unhidden: 15 us of a 30 us 128 x 128 x 128 tile, 21 of 84 us at
    // 256 x 128 x 256 (tools/bench_ci_jobs.py) - a rank-128 update moves 16 B per 256 flop, the chip's HBM balance.
    double *cbp = S + (int64_t)(row0 + wr * (BM / 4) + l4) * ld + col0 + wq * 16 + l15;  // element (mi, ni, r): + (16 mi + 4 r) ld + 32 ni
    auto live_at = [&](int mi, int ni) {
        const int rbase = row0 + wr * (BM / 4) + mi * 16, cbase = col0 + (2 * ni + wq) * 16;
        return rbase < u.r1 && (cbase < Np ? cbase >= (rbase & ~63) : cbase < Np + u.wlim);
    };
    double cb[4][4];
    auto load_batch = [&](int mi) {  // mi may be a run-time value: nothing here indexes a register array with it
        const double *cp = cbp + (int64_t)(16 * mi) * ld;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
#ifdef GPBO_CI_DIAG_NO_CLOAD   // timing only (wrong results): the C tile is not read
            if (false) {
#else
            if (live_at(mi, ni)) {
#endif
#pragma unroll
                for (int r = 0; r < 4; ++r) cb[ni][r] = cp[(int64_t)(4 * r) * ld + 32 * ni];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) cb[ni][r] = 0.0;
            }
        }
    };
    auto fold_batch = [&](auto mi_tag) {  // compile-time mi: a run-time index would send the accumulators to scratch
        constexpr int mi = decltype(mi_tag)::value;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] -= cb[ni][r];
    };
    lds_frag(a0, b0, 0, 0);
    int cur = 0;
    auto k_tile = [&](auto kt_tag) {  // kt_tag::value = the k tile's index while C is on the move (0 .. MI), else -1
        constexpr int KT = decltype(kt_tag)::value;
        const int nxt = (cur == 2) ? 0 : cur + 1;
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a0, b0, 0);
        lds_frag(a1, b1, cur, 4);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a0, b0, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a1, b1, 0);
        lds_frag(a0, b0, cur, 8);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a1, b1, 2);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own share of tile kt+1 has landed (and C batch kt-1)
        __builtin_amdgcn_s_barrier();                      // ... everybody's has; everybody has left tile kt-1
        if constexpr (KT > 0 && KT <= MI) fold_batch(std::integral_constant<int, (KT > 0 ? KT - 1 : 0)>{});  // C batch KT-1
        if constexpr (KT >= 0 && KT < MI) load_batch(KT);                                                    // batch KT on its way
        // tile kt+2 into the stage tile kt-1 occupied; the two waves of a SIMD issue their DMAs half a tile apart
        if (pk < nk && wq == 0) stage_next();
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a0, b0, 0);
        lds_frag(a1, b1, cur, 12);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a0, b0, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a1, b1, 0);
        lds_frag(a0, b0, nxt, 0);  // first step of the next tile (stale but in-bounds after the last one)
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(a1, b1, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (pk < nk && wq != 0) stage_next();
        cur = nxt;
    };
    // the first MI + 1 k tiles also move C (K >= 16 (MI + 1) = 80: every update of the plan has K >= 128)
    k_tile(std::integral_constant<int, 0>{});
    k_tile(std::integral_constant<int, 1>{});
    k_tile(std::integral_constant<int, 2>{});
    if constexpr (MI == 4) {
        k_tile(std::integral_constant<int, 3>{});
        k_tile(std::integral_constant<int, 4>{});
    }
    CI_TSTAMP();
    for (int kt = MI + 1; kt < nk; ++kt) k_tile(std::integral_constant<int, -1>{});
    CI_TSTAMP();
    // out = C - sum = -acc on the live part of the tile (masks are uniform per 16 x 16 MFMA tile: 64-granular)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
#ifdef GPBO_CI_DIAG_NO_STORE   // timing only (wrong results): one lane in 64 stores
            if (live_at(mi, ni) && lane == 0) {
#else
            if (live_at(mi, ni)) {
#endif
                double *cp = cbp + (int64_t)(16 * mi) * ld + 32 * ni;
                // nt: the stacked matrix is far beyond the 256-MiB Infinity Cache (N >= 6144: >= 600 MB) - the hint saves 1.1 % of
                // the factorisation at N = 8192; at N <= 4096 (268 MB) the next update finds its target on chip without it
                if (nt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(-acc[mi][ni][r], cp + (int64_t)(4 * r) * ld);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) cp[(int64_t)(4 * r) * ld] = -acc[mi][ni][r];
                }
            }
        }
    CI_TSTAMP();
#ifdef GPBO_DIAGNOSTICS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    CI_TSTAMP();
#undef CI_TSTAMP
}

// ---------------------------------------------------------------------------------------------------------------
// UPD_SMALL: out[64 x 64] -= sum_k S[k][row0 + m] * S[k][col0 + n].  8 waves as 4 (rows) x 2 (columns): wave tile
// 16 x 32.  Fragments come straight from global memory, 32 k at a time (24 loads in flight per lane), the next 32 k
// are in flight while the current ones are multiplied.
// ---------------------------------------------------------------------------------------------------------------
template <bool HALF>  // HALF: 64 x 32 tile (wave tile 16 x 16), else 64 x 64 (wave tile 16 x 32)
__device__ __forceinline__ void upd_small(double *__restrict__ S, int64_t ld, const CiTile &u) {
    const int row0 = u.row0, col0 = u.col0;
    constexpr int WC = HALF ? 16 : 32;  // columns per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = wid & 3, wcol = wid >> 2;
    const int l15 = lane & 15, l4 = lane >> 4;
    const double *ap = S + (int64_t)(u.k0 + l4) * ld + row0 + wrow * 16 + l15;
    const double *bp = S + (int64_t)(u.k0 + l4) * ld + col0 + wcol * WC + l15;
    double *cp = S + (int64_t)(row0 + wrow * 16 + l4) * ld + col0 + wcol * WC + l15;
    d4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
    double c0[4], c1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        c0[r] = cp[(int64_t)(4 * r) * ld];
        c1[r] = HALF ? 0.0 : cp[(int64_t)(4 * r) * ld + 16];
    }
    if (u.K == 128) {
        // the NEAR launch of every pair (rank 128, on the critical path): ALL fragments are requested at once - one memory
        // round trip instead of three (64 / 96 loads per lane in flight) - and every sum runs as two chains (k step parity)
        double ga[32], gb0[32], gb1[HALF ? 1 : 32];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int64_t o = (int64_t)(4 * s) * ld;
            ga[s] = ap[o];
            gb0[s] = bp[o];
            if (!HALF) gb1[s] = bp[o + 16];
        }
        d4_t acc0b = {0.0, 0.0, 0.0, 0.0}, acc1b = acc0b;
#pragma unroll
        for (int s = 0; s < 32; s += 2) {
            acc0 = mfma_f64_16x16x4(ga[s], gb0[s], acc0);
            acc0b = mfma_f64_16x16x4(ga[s + 1], gb0[s + 1], acc0b);
            if (!HALF) {
                acc1 = mfma_f64_16x16x4(ga[s], gb1[s], acc1);
                acc1b = mfma_f64_16x16x4(ga[s + 1], gb1[s + 1], acc1b);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cp[(int64_t)(4 * r) * ld] = c0[r] - (acc0[r] + acc0b[r]);
            if (!HALF) cp[(int64_t)(4 * r) * ld + 16] = c1[r] - (acc1[r] + acc1b[r]);
        }
        return;
    }
    constexpr int CH = 8;  // k steps (of 4) per chunk
    double fa[2][CH], fb0[2][CH], fb1[2][CH];
    auto load = [&](int set, int kbase) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int64_t o = (int64_t)(kbase + 4 * s) * ld;
            fa[set][s] = ap[o];
            fb0[set][s] = bp[o];
            if (!HALF) fb1[set][s] = bp[o + 16];
        }
    };
    auto mult = [&](int set) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            acc0 = mfma_f64_16x16x4(fa[set][s], fb0[set][s], acc0);
            if (!HALF) acc1 = mfma_f64_16x16x4(fa[set][s], fb1[set][s], acc1);
        }
    };
    const int nch = u.K / (4 * CH);
    load(0, 0);
    for (int c = 0; c < nch; c += 2) {
        if (c + 1 < nch) load(1, (c + 1) * 4 * CH);
        mult(0);
        if (c + 1 < nch) {
            if (c + 2 < nch) load(0, (c + 2) * 4 * CH);
            mult(1);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        cp[(int64_t)(4 * r) * ld] = c0[r] - acc0[r];
        if (!HALF) cp[(int64_t)(4 * r) * ld + 16] = c1[r] - acc1[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Cholesky of one 64 x 64 block AND the inverse of its factor, in one elimination.  The eliminated matrix is the
// 128 x 64 stack [D; I] (Mtop = D, lower triangle, Mbot = I; both [64][LDM] in LDS): the column operations of the
// Cholesky, applied to the identity rows as well, leave L^-T in them.  The 64 columns go 16 at a time:
//   (1) wave 0 eliminates the 16 x 16 diagonal sub-block and the identity rows under it in registers (lane = row,
//       pivots and multipliers broadcast with v_readlane);
//   (2) the three non-zero 16 x 16 blocks of the panel are multiplied by the sub-block's L^-T on the matrix cores;
//   (3) the trailing 16 x 16 blocks are updated on the matrix cores - wave 0 takes the next diagonal sub-block first
//       and goes straight on to (1) while the other seven waves do the rest.
// All 8 waves call it (barriers inside); the caller has synchronised after filling Mtop / Mbot.
// has_bg: the background waves (ci_is_bg_wave) do not help with (3) but run bg(s) there, s = 0, 1, 2 - work of the caller that touches neither Mtop
// nor Mbot and contains no barrier (wave 0's elimination is ~3000 cycles during which they would otherwise wait).
// Returns (wave 0 only) the 1-based column of the first non-positive / non-finite pivot, or 0.
// ---------------------------------------------------------------------------------------------------------------
// DP-DPP helpers of the third form of the sixteen-column phase (eliminate_dpp below): gfx90a+ lets 64-bit VALU operations take
// a DPP operand with ONE control, row_newbcast:k (every lane reads lane k of its row of 16).  Inline assembly: the
// compiler's hazard recogniser does not look inside, so the two wait states "VALU writes a VGPR, DPP reads it" are written
// out (s_nop 1) wherever the producer can be that close.
template <int K>
__device__ __forceinline__ double ci_bcast16(double v) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
    return r;
}
template <int K, bool NOP>
__device__ __forceinline__ void ci_fmac_nbcast(double &acc, double bsrc, double mul) {   // acc -= bsrc[lane K of the row] * mul
    if (NOP)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(mul), "n"(K));
    else
        asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(mul), "n"(K));
}
template <int C, int K>
struct CiDppUpd {   // columns K .. 15 of both 16-row blocks -= column C * L[K][C]
    static __device__ __forceinline__ void run(double (&xt)[16], double (&xb)[16]) {
        if constexpr (K < 16) {
            ci_fmac_nbcast<K, K == C + 1>(xt[K], xt[C], xt[C]);
            ci_fmac_nbcast<K, false>(xb[K], xt[C], xb[C]);
            CiDppUpd<C, K + 1>::run(xt, xb);
        }
    }
};
template <int C>
struct CiDppCol {
    static __device__ __forceinline__ void run(double (&xt)[16], double (&xb)[16], int col0, int &first_bad) {
        if constexpr (C < 16) {
            const double piv = ci_bcast16<C>(xt[C]);
            const bool ok = (piv > 0.0) && (piv < 1.0e300);
            first_bad = (!ok && first_bad == 0) ? col0 + C + 1 : first_bad;
            const double rs = rsqrt_refined(piv);
            xt[C] *= rs;
            xb[C] *= rs;
            CiDppUpd<C, C + 1>::run(xt, xb);
            CiDppCol<C + 1>::run(xt, xb, col0, first_bad);
        }
    }
};

// The waves that carry a caller's background work through an elimination: 1, 5, 6, 7 - none of them on wave 0's SIMD (waves
// w and w + 4 share one; fp64 MFMAs and fp64 vector instructions of a SIMD do not overlap, and wave 0's sixteen-column
// phase is bound by instruction issue: with the X strips on waves 4..7 the second elimination took 30.0k cycles against the
// first one's 22.0k).  Waves 2 and 3 keep the few trailing products; wave 4 does nothing during an elimination.
__device__ __forceinline__ bool ci_is_bg_wave(int w) { return w == 1 || w >= 5; }

template <typename BG>
__device__ __forceinline__ int factor64(double *Mtop, double *Mbot, int w, int lane, bool has_bg, BG &&bg,
                                        unsigned long long *fst = nullptr /* GPBO_CI_F64_STAMPS builds: wave 0's stamps */) {
#ifdef GPBO_CI_F64_STAMPS
    int fn = 0;
#define F64_STAMP() do { if (fst && w == 0 && lane == 0 && fn < 8) fst[fn++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define F64_STAMP() do { } while (0)
#endif
    const int l15 = lane & 15, l4 = lane >> 4;
    auto rowp = [&](int R) { return (R < NB) ? Mtop + R * LDM : Mbot + (R - NB) * LDM; };  // row R of the stack
    int first_bad = 0;
    auto eliminate = [&](int s) {
        const int l31 = lane & 31;
        double *row = ((l31 < PB) ? Mtop + (PB * s + l31) * LDM : Mbot + (PB * s + (l31 - PB)) * LDM) + PB * s;
        double x[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) x[k] = row[k];
#pragma unroll
        for (int c = 0; c < PB; ++c) {
            const double piv = readlane_f64(x[c], c);
            const bool ok = (piv > 0.0) && (piv < 1.0e300);
            first_bad = (!ok && first_bad == 0) ? PB * s + c + 1 : first_bad;
            x[c] *= rsqrt_refined(piv);
            // (issue-bound, ~7 cycles per instruction of this lone wave: starting the next pivot's reciprocal square root right
            //  after the first update and issuing its eight links between the others gained nothing - 24.9k against 23.9k cycles)
            double m[PB];  // the multipliers first (wave-uniform: scalar registers), then the updates: no hazard slots between
#pragma unroll
            for (int k = c + 1; k < PB; ++k) m[k] = readlane_f64(x[c], k);
#pragma unroll
            for (int k = c + 1; k < PB; ++k) x[k] = fma(-x[c], m[k], x[k]);
        }
        if (lane < 2 * PB) {
#pragma unroll
            for (int k = 0; k < PB; ++k) row[k] = (lane < PB && k > lane) ? 0.0 : x[k];
        }
    };
    // Round 4: the same sixteen columns in four MICRO-BLOCKS of four.  The register phase above is bound by instruction issue
    // on this one wave (~650 instructions per sixteen columns at ~7 cycles: 120 column updates of two v_readlane + one fma
    // each, sixteen pivot chains); here only the 6 updates INSIDE a micro-block stay in registers and the other columns of
    // the sixteen get each micro-block as ONE rank-4 product per 16-row block on the matrix core
    // (v_mfma_f64_16x16x4_f64: C[row][col] -= X[row][c0 + k] X[col][c0 + k], k < 4 - the A operand of the diagonal block,
    // negated and masked to the columns still open, IS the B operand).  LDS is the transposer between the two layouts
    // (lane = row for the register phase, the MFMA's fragments for the product); only this wave touches these rows and
    // columns, LDS serves a wave's accesses in order, so no barrier is involved.
    auto eliminate_mb = [&](int s) {
        const int l31 = lane & 31;
        double *row = ((l31 < PB) ? Mtop + (PB * s + l31) * LDM : Mbot + (PB * s + (l31 - PB)) * LDM) + PB * s;
        double *ctop = Mtop + (PB * s + l4) * LDM + PB * s + l15;       // C fragments: rows l4 + 4 r, column l15
        double *cbot = Mbot + (PB * s + l4) * LDM + PB * s + l15;
        const double *atop = Mtop + (PB * s + l15) * LDM + PB * s + l4; // A fragments: row l15, k = l4 (+ c0)
        const double *abot = Mbot + (PB * s + l15) * LDM + PB * s + l4;
#pragma unroll
        for (int mb = 0; mb < PB / 4; ++mb) {
            const int c0 = 4 * mb;
            double x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = row[c0 + k];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double piv = readlane_f64(x[c], c0 + c);
                const bool ok = (piv > 0.0) && (piv < 1.0e300);
                first_bad = (!ok && first_bad == 0) ? PB * s + c0 + c + 1 : first_bad;
                x[c] *= rsqrt_refined(piv);
                double m[4];
#pragma unroll
                for (int k = c + 1; k < 4; ++k) m[k] = readlane_f64(x[c], c0 + k);
#pragma unroll
                for (int k = c + 1; k < 4; ++k) x[k] = fma(-x[c], m[k], x[k]);
            }
            if (lane < 2 * PB) {
#pragma unroll
                for (int k = 0; k < 4; ++k) row[c0 + k] = (lane < PB && c0 + k > lane) ? 0.0 : x[k];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (mb < PB / 4 - 1) {
                const double at = atop[c0], ab = abot[c0];
                const double b = (l15 >= c0 + 4) ? -at : 0.0;   // finished columns (and this micro-block's own) stay as they are
                d4_t ct, cb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ct[r] = ctop[4 * r * LDM];
                    cb[r] = cbot[4 * r * LDM];
                }
                ct = mfma_f64_16x16x4(at, b, ct);
                cb = mfma_f64_16x16x4(ab, b, cb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ctop[4 * r * LDM] = ct[r];
                    cbot[4 * r * LDM] = cb[r];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        // entries above the diagonal of the top block in LATER micro-blocks' columns were carried through the products:
        // zero them, as eliminate() does (nothing reads them, but the factor that is stored must be a triangle)
        if (lane < PB) {
#pragma unroll
            for (int k = 1; k < PB; ++k)
                if (k > lane) row[k] = 0.0;
        }
    };
    // Third form (round 4): both 16-row blocks of a row index in ONE lane's registers (lane l < 16: row l of the diagonal block
    // in xt, identity row l in xb), multipliers by DP-DPP row_newbcast instead of v_readlane pairs: two instructions per
    // column update instead of three, one instead of two for the pivot.  The four rows of 16 lanes compute the same thing.
    auto eliminate_dpp = [&](int s) {
        double *rt = Mtop + (PB * s + l15) * LDM + PB * s, *rb = Mbot + (PB * s + l15) * LDM + PB * s;
        double xt[PB], xb[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) {
            xt[k] = rt[k];
            xb[k] = rb[k];
        }
        int fb = first_bad;
        CiDppCol<0>::run(xt, xb, PB * s, fb);
        first_bad = __builtin_amdgcn_readfirstlane(fb);
        if (lane < PB) {
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                rt[k] = (k > lane) ? 0.0 : xt[k];
                rb[k] = xb[k];
            }
        }
    };
#ifndef GPBO_CI_DPP
#define GPBO_CI_DPP 0   /* measured: 0.2 - 1.0 % faster than the register phase: not worth hand-written hazard slots */
#endif
#ifndef GPBO_CI_MICROBLOCK
#define GPBO_CI_MICROBLOCK 0   /* measured: 2 - 5 % SLOWER than the register phase; tools/build_variant.sh ci_mb cholinv "-DGPBO_CI_MICROBLOCK=1" */
#endif
    auto panel = [&](int R0, int s) {  // rows R0.. (16) of column block s  <-  (those rows) * L_ss^-T
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        const double *ra = rowp(R0 + l15) + PB * s;
#pragma unroll
        for (int kk = 0; kk < PB; kk += 4) {
            const double a = ra[kk + l4];
            const double b = Mbot[(PB * s + kk + l4) * LDM + PB * s + l15];
            acc = mfma_f64_16x16x4(a, b, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) rowp(R0 + l4 + 4 * r)[PB * s + l15] = acc[r];
    };
    auto trail = [&](int R0, int c, int s) {  // rows R0.. of column block c  -=  (rows R0.., block s) * D[block c, block s]^T
        d4_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = rowp(R0 + l4 + 4 * r)[PB * c + l15];
        const double *ra = rowp(R0 + l15) + PB * s;
#pragma unroll
        for (int kk = 0; kk < PB; kk += 4) {
            const double a = -ra[kk + l4];
            const double b = Mtop[(PB * c + l15) * LDM + PB * s + kk + l4];
            acc = mfma_f64_16x16x4(a, b, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) rowp(R0 + l4 + 4 * r)[PB * c + l15] = acc[r];
    };
    constexpr int NS = NB / PB;  // 4
    F64_STAMP();
    if (w == 0) { if (GPBO_CI_DPP) eliminate_dpp(0); else if (GPBO_CI_MICROBLOCK) eliminate_mb(0); else eliminate(0); }
    F64_STAMP();
    __syncthreads();
    F64_STAMP();
    for (int s = 0; s < NS; ++s) {
        if (w < NS - 1) {  // (2): D row blocks s+1..3 and identity-part row blocks 0..s-1 - always three
            const int nA = NS - 1 - s;
            panel(w < nA ? PB * (s + 1 + w) : NB + PB * (w - nA), s);
        }
        if (s == 0) F64_STAMP();
        __syncthreads();
        if (s == 0) F64_STAMP();
        if (s == NS - 1) break;
        if (w == 0) {  // (3)
            trail(PB * (s + 1), s + 1, s);
            // (same wave, LDS in order: no barrier - but the COMPILER must not move the loads below above these stores; it did
            //  exactly that between the phases of eliminate_mb until a memory clobber stood between them)
            asm volatile("" ::: "memory");
            if (s == 0) F64_STAMP();
            if (GPBO_CI_DPP) eliminate_dpp(s + 1); else if (GPBO_CI_MICROBLOCK) eliminate_mb(s + 1); else eliminate(s + 1);
            if (s == 0) F64_STAMP();
        } else if (has_bg && ci_is_bg_wave(w)) {
            bg(s);
        } else {
            // helper waves: 1, 2, 3, 5, 6, 7 - or 2, 3 beside the background waves; wave 4 (wave 0's SIMD) stays idle
            const int nh = has_bg ? 2 : 6;
            const int hw = (w == 4) ? -1 : has_bg ? w - 2 : (w < 4 ? w - 1 : w - 2);   // this wave's rank among them
            int q = 0;
            for (int c = s + 1; c < NS; ++c) {
                for (int R = c; R < NS; ++R) {
                    if (R == s + 1 && c == s + 1) continue;
                    if (q % nh == hw) trail(PB * R, c, s);
                    ++q;
                }
                for (int m = 0; m <= s; ++m) {
                    if (q % nh == hw) trail(NB + PB * m, c, s);
                    ++q;
                }
            }
        }
        __syncthreads();
        if (s == 0) F64_STAMP();
    }
#undef F64_STAMP
    return first_bad;
}

// ---------------------------------------------------------------------------------------------------------------
// PAIR(p), column tile pt: the 128 rows [128 p, 128 p + 128) in ONE workgroup launch - what used to be four launches
// (diagonal block, narrow update, diagonal block, row scaling).  Every workgroup of the launch factorises the pair's
// 128 x 128 diagonal block D = [D11 A12; . D22] redundantly (identical arithmetic, identical bits):
//     L11, inv(L11)  <-  D11                         factor64
//     R12 = inv(L11) A12                             (= L21^T)
//     D22 -= R12^T R12 ;  L22, inv(L22)  <-  D22     factor64
// and then applies inv(L) = [inv(L11) 0; -inv(L22) L21 inv(L11)  inv(L22)] to its own 64 columns X = [X1; X2] of the
// two block rows as three products that never leave the registers:
//     X1' = inv(L11) X1 ;   X2 <- X2 - R12^T X1' ;   X2' = inv(L22) X2
// (waves 1, 5, 6, 7, one 16-column strip each: the fp64 16x16x4 MFMA's C layout - row = lane/16 + 4 r - IS the B operand
// layout of the next product, k = lane/16 within k step r; the first two products run beside the second elimination).  W's diagonal block of the pair is the same computation
// on X = I.  The pair's diagonal block of A is left alone: nothing reads it later, other workgroups are reading it now.
// LDS: four [64][LDM] blocks - B0: D11 -> L11, then D22 -> L22;  B1: I -> inv(L11)^T (kept);  B2: I -> inv(L22)^T;
// B3: A12 -> R12.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pair_body(double *__restrict__ S, int64_t ld, int Np, int p, int pt,
                                          int32_t *__restrict__ info, double *smem, unsigned long long *stamps) {
    double *B0 = smem, *B1 = smem + NB * LDM, *B2 = smem + 2 * NB * LDM, *B3 = smem + 3 * NB * LDM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int r0 = 128 * p;
#ifdef GPBO_DIAGNOSTICS
    int nst = 0;
#ifdef GPBO_CI_F64_STAMPS
#define CI_STAMP() do { (void)nst; } while (0)
#else
#define CI_STAMP() do { if (stamps && pt == 0 && tid == 0) stamps[nst++] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#else
#define CI_STAMP() do { } while (0)
#endif
    CI_STAMP();
    // column tile: A part beyond the pair's diagonal block, then W part incl. the pair's own diagonal block of W
    const int nA = (Np - (r0 + 128)) / 64;
    const int c0 = (pt < nA) ? r0 + 128 + 64 * pt : Np + 64 * (pt - nA);
    const int wid_blk = (c0 >= Np) ? (c0 - Np - r0) / 64 : -1;  // 0 / 1: column block of W's diagonal 128 x 128 block
    const bool ident = c0 >= Np && c0 - Np >= r0;
    const double *Dg = S + (int64_t)r0 * ld + r0;

    // Everything the workgroup reads from global memory is requested here, in the order it is needed: D11 and A12
    // (eight elements per thread each: registers first, LDS after - one round trip, not eight), the wave's tiles of D22
    // in MFMA C layout, and the X strip of waves 1, 5, 6, 7 as B-operand fragments (consumed after the first elimination).
    double dv[8], av[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + 512 * i, r = e >> 6, c = e & 63;
        dv[i] = Dg[(int64_t)r * ld + c];
        av[i] = Dg[(int64_t)r * ld + 64 + c];
    }
    // Schur step tiles: the ten lower 16 x 16 tiles of D22, tile w for every wave, tiles 8 / 9 for waves 0 / 1 too
    auto schur_tile = [](int i, int *mi, int *ni) {  // i -> (mi, ni), mi >= ni, row-major over the lower triangle
        int m = 0;
        while ((m + 1) * (m + 2) / 2 <= i) ++m;
        *mi = m;
        *ni = i - m * (m + 1) / 2;
    };
    d4_t d22[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int mi, ni;
        schur_tile(q ? 8 + (w & 1) : w, &mi, &ni);
#pragma unroll
        for (int r = 0; r < 4; ++r) d22[q][r] = Dg[(int64_t)(64 + 16 * mi + l4 + 4 * r) * ld + 64 + 16 * ni + l15];
    }
    const bool xw = ci_is_bg_wave(w);   // the X waves: 1, 5, 6, 7
    const int xs = (w == 1) ? 0 : w - 4;  // their strips: 0, 1, 2, 3
    double x1[16], x2[16];
    if (xw) {
        const double *xp = S + (int64_t)(r0 + l4) * ld + c0 + 16 * xs + l15;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (!ident) {
                x1[s] = xp[(int64_t)(4 * s) * ld];
                x2[s] = xp[(int64_t)(64 + 4 * s) * ld];
            } else {  // X = [I; 0] (column block 0) or [0; I] (column block 1): row 4 s + l4, column 16 xs + l15
                const double d = (4 * s + l4 == 16 * xs + l15) ? 1.0 : 0.0;
                x1[s] = (wid_blk == 0) ? d : 0.0;
                x2[s] = (wid_blk == 1) ? d : 0.0;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + 512 * i, r = e >> 6, c = e & 63;
        B0[r * LDM + c] = (c <= r) ? dv[i] : 0.0;
        B1[r * LDM + c] = (c == r) ? 1.0 : 0.0;
        B2[r * LDM + c] = (c == r) ? 1.0 : 0.0;
        B3[r * LDM + c] = av[i];
    }
    __syncthreads();
    CI_STAMP();
#if defined(GPBO_CI_F64_STAMPS) && GPBO_CI_F64_STAMPS == 1   // (= 2: the stamps of the SECOND elimination instead)
    const int bad1 = factor64(B0, B1, w, lane, false, [](int) {}, (stamps && pt == 0) ? stamps : nullptr);
#else
    const int bad1 = factor64(B0, B1, w, lane, false, [](int) {});   // ends with a barrier after the last panel step
#endif
    CI_STAMP();
    // R12 = inv(L11) A12: inv(L11)[r][k] = B1[k][r], zero for k > r.  Wave w: column tile w & 3 of the row tiles
    // {w >> 2, 3 - (w >> 2)} (4 (mi + 1) k steps each: 20 per wave); read completely before the barrier, written after.
    {
        const int ni = w & 3, mia = w >> 2, mib = 3 - mia;
        // a dependent fp64 MFMA chain issues one instruction per ~2 pipe slots: every tile's sum is split over the parity
        // of the k step (two chains per tile, four per wave), added at the end
        d4_t acca = {0.0, 0.0, 0.0, 0.0}, accb = acca, acca2 = acca, accb2 = acca;
        // fully unrolled with wave-uniform guards (mib > mia): the LDS reads of all k steps are in flight at once
#pragma unroll
        for (int ks = 0; ks < 16; ks += 2) {
            if (ks < 4 * (mib + 1)) {
                const int k = 4 * ks + l4;
                const double bv = B3[k * LDM + 16 * ni + l15], bw = B3[(k + 4) * LDM + 16 * ni + l15];
                accb = mfma_f64_16x16x4(B1[k * LDM + 16 * mib + l15], bv, accb);
                accb2 = mfma_f64_16x16x4(B1[(k + 4) * LDM + 16 * mib + l15], bw, accb2);
                if (ks < 4 * (mia + 1)) {
                    acca = mfma_f64_16x16x4(B1[k * LDM + 16 * mia + l15], bv, acca);
                    acca2 = mfma_f64_16x16x4(B1[(k + 4) * LDM + 16 * mia + l15], bw, acca2);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acca[r] += acca2[r];
            accb[r] += accb2[r];
        }
        CI_STAMP();
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            B3[(16 * mia + l4 + 4 * r) * LDM + 16 * ni + l15] = acca[r];
            B3[(16 * mib + l4 + 4 * r) * LDM + 16 * ni + l15] = accb[r];
        }
        __syncthreads();
        CI_STAMP();
    }
    // D22 -= R12^T R12, lower tiles only -> B0 (the elimination reads nothing above the diagonal); B2 = I already
    {
        int mi0, ni0, mi1, ni1;
        schur_tile(w, &mi0, &ni0);
        schur_tile(8 + (w & 1), &mi1, &ni1);
        const bool two = w < 2;
        d4_t acc0 = d22[0], acc1 = d22[1], acc0b = {0.0, 0.0, 0.0, 0.0}, acc1b = acc0b;
#pragma unroll
        for (int ks = 0; ks < 16; ks += 2) {
            const int k = 4 * ks + l4;
            acc0 = mfma_f64_16x16x4(-B3[k * LDM + 16 * mi0 + l15], B3[k * LDM + 16 * ni0 + l15], acc0);
            acc0b = mfma_f64_16x16x4(-B3[(k + 4) * LDM + 16 * mi0 + l15], B3[(k + 4) * LDM + 16 * ni0 + l15], acc0b);
            if (two) {
                acc1 = mfma_f64_16x16x4(-B3[k * LDM + 16 * mi1 + l15], B3[k * LDM + 16 * ni1 + l15], acc1);
                acc1b = mfma_f64_16x16x4(-B3[(k + 4) * LDM + 16 * mi1 + l15], B3[(k + 4) * LDM + 16 * ni1 + l15], acc1b);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc0[r] += acc0b[r];
            acc1[r] += acc1b[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = 16 * mi0 + l4 + 4 * r, cc = 16 * ni0 + l15;
            B0[rr * LDM + cc] = (cc <= rr) ? acc0[r] : 0.0;
            if (two) {
                const int r2 = 16 * mi1 + l4 + 4 * r, c2 = 16 * ni1 + l15;
                B0[r2 * LDM + c2] = (c2 <= r2) ? acc1[r] : 0.0;
            }
        }
        // the strictly upper tiles of B0 still hold L11's zeros / entries: above the diagonal nothing is read
    }
    __syncthreads();
    CI_STAMP();
    // Second elimination, and beside it (waves 1, 5, 6, 7, while wave 0 eliminates): X1' = inv(L11) X1, then
    // X2 <- X2 - R12^T X1' in two halves - both only need B1 and B3, which the elimination does not touch.
    double y1[16];
    double *xp = S + (int64_t)(r0 + l4) * ld + c0 + 16 * xs + l15;
    auto x_bg = [&](int s) {
        if (s == 0) {
            d4_t acc[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)  // the four row tiles are independent chains: k outside, tiles inside
#pragma unroll
                for (int mi = ks / 4; mi < 4; ++mi)
                    acc[mi] = mfma_f64_16x16x4(B1[(4 * ks + l4) * LDM + 16 * mi + l15], x1[ks], acc[mi]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) y1[4 * mi + r] = acc[mi][r];
#pragma unroll
            for (int q = 0; q < 16; ++q) xp[(int64_t)(4 * q) * ld] = y1[q];
        } else {
            const int m0 = 2 * (s - 1);  // s = 1: row tiles 0, 1;  s = 2: row tiles 2, 3
            d4_t acc[2], accb[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
            if (s == 1) {
                acc[0] = d4_t{x2[0], x2[1], x2[2], x2[3]};
                acc[1] = d4_t{x2[4], x2[5], x2[6], x2[7]};
            } else {
                acc[0] = d4_t{x2[8], x2[9], x2[10], x2[11]};
                acc[1] = d4_t{x2[12], x2[13], x2[14], x2[15]};
            }
#pragma unroll
            for (int ks = 0; ks < 16; ks += 2)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    acc[h] = mfma_f64_16x16x4(-B3[(4 * ks + l4) * LDM + 16 * (m0 + h) + l15], y1[ks], acc[h]);
                    accb[h] = mfma_f64_16x16x4(-B3[(4 * ks + 4 + l4) * LDM + 16 * (m0 + h) + l15], y1[ks + 1], accb[h]);
                }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[h][r] += accb[h][r];
            if (s == 1) {
                x2[0] = acc[0][0]; x2[1] = acc[0][1]; x2[2] = acc[0][2]; x2[3] = acc[0][3];
                x2[4] = acc[1][0]; x2[5] = acc[1][1]; x2[6] = acc[1][2]; x2[7] = acc[1][3];
            } else {
                x2[8] = acc[0][0]; x2[9] = acc[0][1]; x2[10] = acc[0][2]; x2[11] = acc[0][3];
                x2[12] = acc[1][0]; x2[13] = acc[1][1]; x2[14] = acc[1][2]; x2[15] = acc[1][3];
            }
        }
    };
#if defined(GPBO_CI_F64_STAMPS) && GPBO_CI_F64_STAMPS == 2
    const int bad2 = factor64(B0, B2, w, lane, true, x_bg, (stamps && pt == 0) ? stamps : nullptr);
#else
    const int bad2 = factor64(B0, B2, w, lane, true, x_bg);
#endif
    CI_STAMP();
    if (pt == 0 && tid == 0) {
        const int bad = bad1 ? r0 + bad1 : (bad2 ? r0 + 64 + bad2 : 0);
        if (bad) atomicCAS(info, 0, bad);
    }
    if (xw) {  // X2' = inv(L22) X2
        d4_t acc[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
#pragma unroll
            for (int mi = ks / 4; mi < 4; ++mi)
                acc[mi] = mfma_f64_16x16x4(B2[(4 * ks + l4) * LDM + 16 * mi + l15], x2[ks], acc[mi]);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) xp[(int64_t)(64 + 16 * mi + 4 * r) * ld] = acc[mi][r];
    }
    CI_STAMP();
#undef CI_STAMP
}

// WITH256: the build that also holds the 256 x 128 tile engine (plans with far_kind = CI_UPD_BIG256 only - a measured null
// result kept reproducible).  Its 128 accumulator registers push the WHOLE kernel to 256 VGPRs with spills and a scratch
// segment; without it (every default plan) the kernel needs 225 registers and no scratch, and the same launches run 1.4 %
// (N = 4096) / 2.1 % (N = 8192) faster.
// The arguments are passed one by one, in the order the critical workgroups need them (PAIR: S, ld, Np, pair, npair; NEAR:
// those and its eight shape words): build.sh compiles this file with -amdgpu-kernarg-preload-count=16, so that the first 16
// dwords arrive in scalar registers WITH the wave instead of through a load from the argument buffer that every workgroup
// used to wait for before it could form its first address.
template <bool WITH256>
__global__ __launch_bounds__(512) void cholinv_kernel(double *S_, int64_t ld_, int32_t Np_, int32_t pair_, int32_t npair_,
                                                      int32_t n_on, int32_t n_k0, int32_t n_K, int32_t n_row0, int32_t n_wlim,
                                                      int32_t n_w, int32_t n_n0, int32_t n_r1, int32_t tile0_, int32_t ntile_,
                                                      int32_t group_, const CiTile *tab_, int32_t *info_,
                                                      unsigned long long *stamps_) {
    CiArgs a;
    a.S = S_;
    a.ld = ld_;
    a.Np = Np_;
    a.info = info_;
    a.stamps = stamps_;
    a.tab = tab_;
    a.l = CiLaunch{pair_, npair_, tile0_, ntile_, group_};
    a.near = CiNear{n_on, n_k0, n_K, n_row0, n_wlim, n_w, n_n0, n_r1};
    __shared__ double smem[SMEM_D];
    const int b = blockIdx.x;
    if (b < a.l.npair) {
        pair_body(a.S, a.ld, a.Np, a.l.pair, b, a.info, smem, a.stamps);
        return;
    }
    if (a.near.on) {  // (npair == 0, group == 1)
        const int hi = b >= a.near.n0;
        const int r = a.near.row0 + 64 * hi;
        const CiTile t = {CI_UPD_SMALL, a.near.k0, a.near.K, r, r + a.near.w * (b - (hi ? a.near.n0 : 0)), a.near.r1, a.near.wlim, a.near.w};
        if (t.w == 32) upd_small<true>(a.S, a.ld, t);
        else upd_small<false>(a.S, a.ld, t);
        return;
    }
    const int first = (b - a.l.npair) * a.l.group;
    const int last = (first + a.l.group < a.l.ntile) ? first + a.l.group : a.l.ntile;
    for (int i = first; i < last; ++i) {
        if (i > first) __syncthreads();  // the previous tile's last LDS reads are done before the ring is refilled
        const CiTile t = a.tab[a.l.tile0 + i];  // wave-uniform: scalar loads
        unsigned long long *ts = nullptr;
#ifdef GPBO_DIAGNOSTICS
        if (a.stamps && a.l.npair == 0) ts = a.stamps + 8 * (i & 1023);  // update-only launches: every tile
#endif
        if (t.kind == CI_UPD_SMALL) { if (t.w == 32) upd_small<true>(a.S, a.ld, t); else upd_small<false>(a.S, a.ld, t); }
        else if (t.kind == CI_UPD_BIG) upd_big<2>(a.S, a.ld, a.Np, t, smem, ts);
        else if (WITH256 && t.kind == CI_UPD_BIG256) upd_big<4>(a.S, a.ld, a.Np, t, smem, ts);
    }
}

// U = W^T restricted to the upper triangle (W = the right half of S, lower triangular): 64x64 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_w_kernel(const double *__restrict__ W, int64_t ldw,
                                                          double *__restrict__ U, int64_t Np) {
    __shared__ double tile[NB * (NB + 1)];
    const int bi = blockIdx.y, bj = blockIdx.x;  // output tile (bi, bj) of U = input tile (bj, bi) of W
    double *Ub = U + ((int64_t)bi * NB) * Np + (int64_t)bj * NB;
    if (bj < bi) {
        for (int e = threadIdx.x; e < NB * NB; e += 256) Ub[(int64_t)(e >> 6) * Np + (e & 63)] = 0.0;
        return;
    }
    const double *Wb = W + ((int64_t)bj * NB) * ldw + (int64_t)bi * NB;
    for (int e = threadIdx.x; e < NB * NB; e += 256)
        tile[(e >> 6) * (NB + 1) + (e & 63)] = Wb[(int64_t)(e >> 6) * ldw + (e & 63)];
    __syncthreads();
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        double v = tile[c * (NB + 1) + r];
        if (bi == bj && c < r) v = 0.0;
        Ub[(int64_t)r * Np + c] = v;
    }
}

}  // namespace

#define CI_FLAT_ARGS(a) (a).S, (a).ld, (a).Np, (a).l.pair, (a).l.npair, (a).near.on, (a).near.k0, (a).near.K, (a).near.row0, (a).near.wlim, \
    (a).near.w, (a).near.n0, (a).near.r1, (a).l.tile0, (a).l.ntile, (a).l.group, (a).tab, (a).info, (a).stamps

// ---- host side ---------------------------------------------------------------------------------------------------
// The plan of a size is built once and its tile table kept on the device (per device, size and options), for the life of
// the process.  Not a capturable operation the FIRST time a size is seen (synchronous upload); afterwards the call only
// enqueues kernels.
struct DevPlan {
    CiPlan plan;
    CiTile *dtiles = nullptr;
    std::vector<CiNear> near;  // per launch; on = 0 where the tiles are not one regular 64 x w update
    bool has256 = false;       // some tile is CI_UPD_BIG256: launch the kernel build that holds that engine
};

// the closed form the kernel uses, checked against the plan's own tiles: any difference and the launch reads the table
static CiNear near_of(const CiPlan &P, const CiLaunch &l) {
    CiNear z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l.npair != 0 || l.ntile <= 0 || l.group != 1) return z;
    const CiTile *T = P.tiles.data() + l.tile0;
    if (T[0].kind != CI_UPD_SMALL || (T[0].w != 32 && T[0].w != 64)) return z;
    CiNear n = {1, T[0].k0, T[0].K, T[0].row0, T[0].wlim, T[0].w, 0, T[0].r1};
    while (n.n0 < l.ntile && T[n.n0].row0 == n.row0) ++n.n0;
    for (int i = 0; i < l.ntile; ++i) {
        const int hi = i >= n.n0, r = n.row0 + 64 * hi;
        const CiTile e = {CI_UPD_SMALL, n.k0, n.K, r, r + n.w * (i - (hi ? n.n0 : 0)), n.r1, n.wlim, n.w};
        const CiTile &t = T[i];
        if (t.kind != e.kind || t.k0 != e.k0 || t.K != e.K || t.row0 != e.row0 || t.col0 != e.col0 || t.r1 != e.r1 ||
            t.wlim != e.wlim || t.w != e.w)
            return z;
    }
    return n;
}

static const DevPlan *plan_for(int Np, const CiPlanOptions &o) {
    static std::mutex mu;
    static std::map<std::array<int, 8>, DevPlan *> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const std::array<int, 8> key = {dev, Np, o.win, o.far_k, o.far_kind, o.defer, o.group_from, o.small_w};
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    DevPlan *dp = new DevPlan;
    dp->plan = ci_plan(Np, o);
    for (const CiLaunch &l : dp->plan.launches) dp->near.push_back(near_of(dp->plan, l));
    for (const CiTile &t : dp->plan.tiles) dp->has256 = dp->has256 || t.kind == CI_UPD_BIG256;
    const size_t bytes = dp->plan.tiles.size() * sizeof(CiTile);
    if (bytes) {
        if (hipMalloc(&dp->dtiles, bytes) != hipSuccess) { delete dp; return nullptr; }
        if (hipMemcpy(dp->dtiles, dp->plan.tiles.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(dp->dtiles);
            delete dp;
            return nullptr;
        }
    }
    cache[key] = dp;
    return dp;
}

static bool ci_sizes_ok(const double *S, int64_t ld, int64_t Np) {
    // 16-byte alignment: upd_big fills LDS with 16-byte global_load_lds straight from S (ld even keeps every row aligned)
    return S && ((uintptr_t)S & 15) == 0 && Np >= 128 && Np % 128 == 0 && ld >= 2 * Np && (ld & 1) == 0 &&
           Np <= GPBO_CHOLINV_MAX_NP && (int64_t)15 * ld + 256 <= 0x7fffffffLL;
}

// S: [Np x ld] row-major, ld >= 2 Np, columns [0, Np) = the symmetric positive definite matrix, [Np, 2 Np) = zeros.
// On return columns [Np, 2 Np) hold inv(L) (lower triangular); the upper block triangle of [0, Np) holds L^T except its
// 128 x 128 diagonal blocks.  *info (cleared by the caller on this stream) receives the 1-based index of the first bad
// pivot.  opt: NULL or int32[7] {win, far_k, far_kind, defer + 1, max_launches, group_from + 1, small_w}, 0 = default.
int gpbo_cholinv_run(double *S, int64_t ld, int64_t Np, int32_t *info, const int *opt, hipStream_t st) {
    if (!ci_sizes_ok(S, ld, Np) || !info) return GPBO_ERR_ARG;
    const CiPlanOptions o = ci_options_from((int)Np, opt);
    if (!ci_options_ok(o)) return GPBO_ERR_ARG;
    const DevPlan *dp = plan_for((int)Np, o);
    if (!dp) return GPBO_ERR_LAUNCH;
    CiArgs a;
    a.S = S;
    a.ld = ld;
    a.Np = (int32_t)Np;
    a.info = info;
    a.stamps = nullptr;
    a.tab = dp->dtiles;
    int left = (opt && opt[4] > 0) ? opt[4] : (int)dp->plan.launches.size();  // tests: stop after this many launches
#ifdef GPBO_DIAGNOSTICS
    // GPBO_CI_STAMPS=1 (timing builds only): cycle stamps of workgroup 0 of every PAIR launch, averaged and printed
    static unsigned long long *dstamps = nullptr;
    const bool want_stamps = getenv("GPBO_CI_STAMPS") && atoi(getenv("GPBO_CI_STAMPS"));
    if (want_stamps && !dstamps && hipMalloc(&dstamps, 8 * 8 * 1024) != hipSuccess) return GPBO_ERR_LAUNCH;
    int npair = 0;
#endif
#ifdef GPBO_DIAGNOSTICS
    // GPBO_CI_CEILING (timing builds only, WRONG results): what ANY scheduler of these tiles could reach.
    //   1: the critical path alone - the NEAR and PAIR launches without their filler tiles;
    //   2: the work alone - every update tile of the plan in ONE launch, no dependencies, no PAIR workgroups.
    static const int ceiling = getenv("GPBO_CI_CEILING") ? atoi(getenv("GPBO_CI_CEILING")) : 0;
    if (ceiling == 2) {
        a.near = CiNear{0, 0, 0, 0, 0, 0, 0, 0};
        a.l = CiLaunch{-1, 0, 0, (int)dp->plan.tiles.size(), 1};
        if (dp->has256) hipLaunchKernelGGL(cholinv_kernel<true>, dim3((unsigned)a.l.ntile), dim3(512), 0, st, CI_FLAT_ARGS(a));
        else hipLaunchKernelGGL(cholinv_kernel<false>, dim3((unsigned)a.l.ntile), dim3(512), 0, st, CI_FLAT_ARGS(a));
        GPBO_CHECK_LAUNCH();
        return GPBO_OK;
    }
#endif
    size_t li = 0;
    for (const CiLaunch &l0 : dp->plan.launches) {
        CiLaunch l = l0;
        a.near = dp->near[li++];
        if (left-- <= 0) break;
#ifdef GPBO_DIAGNOSTICS
        if (ceiling == 1 && l.npair > 0) l.ntile = 0;
#endif
        a.l = l;
        const int nblk = ci_launch_blocks(l);
        if (nblk <= 0) continue;
#ifdef GPBO_DIAGNOSTICS
        a.stamps = (want_stamps && l.npair > 0 && npair < 1024) ? dstamps + 8 * npair++ : nullptr;
#endif
        if (dp->has256) hipLaunchKernelGGL(cholinv_kernel<true>, dim3((unsigned)nblk), dim3(512), 0, st, CI_FLAT_ARGS(a));
        else hipLaunchKernelGGL(cholinv_kernel<false>, dim3((unsigned)nblk), dim3(512), 0, st, CI_FLAT_ARGS(a));
    }
    GPBO_CHECK_LAUNCH();
#ifdef GPBO_DIAGNOSTICS
    if (want_stamps && npair > 0) {
        std::vector<unsigned long long> h(8 * npair);
        if (hipStreamSynchronize(st) != hipSuccess) return GPBO_ERR_LAUNCH;
        if (hipMemcpy(h.data(), dstamps, 8 * 8 * npair, hipMemcpyDeviceToHost) != hipSuccess) return GPBO_ERR_LAUNCH;
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < npair; ++i)
            for (int q = 0; q < 7; ++q) sum[q] += (double)(h[8 * i + q + 1] - h[8 * i + q]);
        fprintf(stderr, "PAIR stamps (wave 0, cycles, mean of %d): load %.0f  factor1 %.0f  r12 products %.0f  r12 barrier+store %.0f  schur %.0f  "
                "factor2 (+X1', X2 update) %.0f  end %.0f\n", npair, sum[0] / npair, sum[1] / npair, sum[2] / npair, sum[3] / npair,
                sum[4] / npair, sum[5] / npair, sum[6] / npair);
    }
#endif
    return GPBO_OK;
}

int gpbo_launch_transpose_w(const double *W, int64_t ldw, int64_t Np, double *U, hipStream_t st) {
    dim3 grid((unsigned)(Np / NB), (unsigned)(Np / NB));
    hipLaunchKernelGGL(transpose_w_kernel, grid, dim3(256), 0, st, W, ldw, U, Np);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

// ---- plan introspection for the CPU simulator (no GPU needed) -----------------------------------------------------
extern "C" int gpbo_cholinv_plan(int64_t Np, const int32_t *opt, int64_t *n_launch, int64_t *n_tile, int32_t *launches,
                                 int32_t *tiles) {
    if (Np < 128 || Np % 128 || Np > GPBO_CHOLINV_MAX_NP || !n_launch || !n_tile) return GPBO_ERR_ARG;
    const CiPlanOptions o = ci_options_from((int)Np, opt);
    if (!ci_options_ok(o)) return GPBO_ERR_ARG;
    const CiPlan P = ci_plan((int)Np, o);
    if (launches) {
        if (*n_launch < (int64_t)P.launches.size()) return GPBO_ERR_WORKSPACE;
        for (size_t i = 0; i < P.launches.size(); ++i) {
            const CiLaunch &l = P.launches[i];
            const int32_t v[5] = {l.pair, l.npair, l.tile0, l.ntile, l.group};
            for (int q = 0; q < 5; ++q) launches[5 * i + q] = v[q];
        }
    }
    if (tiles) {
        if (*n_tile < (int64_t)P.tiles.size()) return GPBO_ERR_WORKSPACE;
        for (size_t i = 0; i < P.tiles.size(); ++i) {
            const CiTile &t = P.tiles[i];
            const int32_t v[8] = {t.kind, t.k0, t.K, t.row0, t.col0, t.r1, t.wlim, t.w};
            for (int q = 0; q < 8; ++q) tiles[8 * i + q] = v[q];
        }
    }
    *n_launch = (int64_t)P.launches.size();
    *n_tile = (int64_t)P.tiles.size();
    return GPBO_OK;
}

// The factorisation alone on a caller-provided stacked matrix (tests, tools/bench_factorise.py).
extern "C" int gpbo_cholinv_f64(double *S, int64_t ld, int64_t Np, int32_t *info, const int32_t *opt, void *stream) {
    hipStream_t st = gpbo_stream(stream);
    if (!ci_sizes_ok(S, ld, Np) || !info) return GPBO_ERR_ARG;
    if (hipMemsetAsync(info, 0, sizeof(int32_t), st) != hipSuccess) return GPBO_ERR_LAUNCH;
    return gpbo_cholinv_run(S, ld, Np, info, opt, st);
}

// One launch made of the PAIR workgroups of `pair` (pair < 0: none) and the given tiles, `reps` times (tests: a tile kind
// against NumPy; tools/bench_ci_jobs.py: its time).  tiles: host array, 8 words per tile as gpbo_cholinv_plan writes them.
// Synchronous (uploads the tiles, waits for the launches).
extern "C" int gpbo_cholinv_tiles_f64(double *S, int64_t ld, int64_t Np, int32_t *info, int32_t pair,
                                      const int32_t *tiles, int64_t ntile, int32_t group, int32_t reps, void *stream) {
    if (!ci_sizes_ok(S, ld, Np) || !info || reps < 1 || group < 1 || group > 64 || ntile < 0 || (ntile > 0 && !tiles) || ntile > (1 << 24)) return GPBO_ERR_ARG;
    if (pair >= 0 && 128 * ((int64_t)pair + 1) > Np) return GPBO_ERR_ARG;
    if (pair < 0 && ntile == 0) return GPBO_ERR_ARG;
    std::vector<CiTile> h((size_t)ntile);
    for (int64_t i = 0; i < ntile; ++i) {
        const int32_t *v = tiles + 8 * i;
        CiTile t = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
        const int th = t.kind == CI_UPD_SMALL ? 64 : t.kind == CI_UPD_BIG ? 128 : t.kind == CI_UPD_BIG256 ? 256 : 0;
        if (t.kind == CI_UPD_SMALL && t.w != 0 && t.w != 32 && t.w != 64) return GPBO_ERR_ARG;
        const int tw = t.kind == CI_UPD_SMALL ? (t.w == 32 ? 32 : 64) : 128;
        if (!th || t.K < (th == 64 ? 32 : 128) || t.K % 32 || t.k0 < 0 || t.k0 + t.K > t.row0 || t.row0 % 64 || t.col0 % tw ||
            t.row0 >= Np || t.col0 < 0 || t.col0 + tw > 2 * Np || t.r1 > Np || t.r1 <= t.row0 || t.wlim < 0 || t.wlim > Np ||
            (th == 64 && (t.col0 < t.row0 || t.col0 + tw > Np + t.wlim || t.row0 + 64 > t.r1)))
            return GPBO_ERR_ARG;
        h[(size_t)i] = t;
    }
    CiTile *d = nullptr;
    if (ntile && (hipMalloc(&d, (size_t)ntile * sizeof(CiTile)) != hipSuccess)) return GPBO_ERR_LAUNCH;
    if (ntile && hipMemcpy(d, h.data(), (size_t)ntile * sizeof(CiTile), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return GPBO_ERR_LAUNCH;
    }
    CiArgs a;
    a.S = S;
    a.ld = ld;
    a.Np = (int32_t)Np;
    a.info = info;
    a.stamps = nullptr;
    a.tab = d;
    a.l = CiLaunch{pair, pair >= 0 ? ci_pair_ntiles((int)Np) : 0, 0, (int32_t)ntile, group};
    a.near = CiNear{0, 0, 0, 0, 0, 0, 0, 0};
#ifdef GPBO_DIAGNOSTICS
    unsigned long long *dst = nullptr;
    const bool want_stamps = getenv("GPBO_CI_STAMPS") && atoi(getenv("GPBO_CI_STAMPS")) && pair < 0;
    if (want_stamps && hipMalloc(&dst, 8 * 8 * 1024) == hipSuccess) {
        (void)hipMemset(dst, 0, 8 * 8 * 1024);
        a.stamps = dst;
    }
#endif
    bool any256 = false;
    for (const CiTile &t : h) any256 = any256 || t.kind == CI_UPD_BIG256;
    for (int r = 0; r < reps; ++r) {
        if (any256) hipLaunchKernelGGL(cholinv_kernel<true>, dim3((unsigned)ci_launch_blocks(a.l)), dim3(512), 0, gpbo_stream(stream), CI_FLAT_ARGS(a));
        else hipLaunchKernelGGL(cholinv_kernel<false>, dim3((unsigned)ci_launch_blocks(a.l)), dim3(512), 0, gpbo_stream(stream), CI_FLAT_ARGS(a));
    }
    const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(gpbo_stream(stream)) == hipSuccess;
#ifdef GPBO_DIAGNOSTICS
    if (dst) {
        std::vector<unsigned long long> h(8 * 1024);
        if (hipMemcpy(h.data(), dst, 8 * 8 * 1024, hipMemcpyDeviceToHost) == hipSuccess) {
            const int n = ntile < 1024 ? (int)ntile : 1024;
            double sum[5] = {0, 0, 0, 0, 0};
            unsigned long long first = ~0ull, lastt = 0;
            for (int i = 0; i < n; ++i) {
                for (int q = 0; q < 5; ++q) sum[q] += (double)(h[8 * i + q + 1] - h[8 * i + q]);
                if (h[8 * i] < first) first = h[8 * i];
                if (h[8 * i + 5] > lastt) lastt = h[8 * i + 5];
            }
            fprintf(stderr, "tile stamps (cycles, mean of %d tiles of the last launch): prologue %.0f  k tiles with C %.0f  steady k tiles %.0f  "
                    "stores issued %.0f  stores done %.0f;  first start -> last end %.0f\n", n, sum[0] / n, sum[1] / n, sum[2] / n,
                    sum[3] / n, sum[4] / n, (double)(lastt - first));
        }
        (void)hipFree(dst);
    }
#endif
    if (d) (void)hipFree(d);
    return ok ? GPBO_OK : GPBO_ERR_LAUNCH;
}
