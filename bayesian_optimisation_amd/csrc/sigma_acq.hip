// Posterior variance + acquisition + arg-max, fused.
//
// Replaces, for every candidate c,
//     Sigma = K** - K*x inv K*x^T ; cov_func = sqrt(|diag Sigma|)        point_selector.py:91,98
//     acq   = explore * cov_func - mean_func ; argwhere(acq == amax)[0]   point_selector.py:204,207
// The reference forms the full M x M matrix; only its diagonal is ever used, and
//     diag_c = prior_var - |v_c|^2,   v_c = L^-1 k_c = U^T k_c,   U = L^-T (upper triangular).
// V = KsT^T * U is a dense triangular product: it runs on v_mfma_f64_16x16x4_f64 and V itself is
// never stored - each 256 x 128 block of V is squared and row-summed out of the accumulators.
//
// Workgroup = 512 threads = 8 waves, owns 256 candidates end to end:
//   for each 128-column block jb of V:   k runs over [0, 128*(jb+1))  (U is upper triangular)
//       16-deep k tiles: KsT tile [16 x 256] and U tile [16 x 128] go global -> LDS directly (global_load_lds,
//       three-stage ring, one barrier in the middle of each k tile), MFMA 4x4 tiles of 16x16 per wave;
//       inside the diagonal block, 16x16 tiles of U that lie wholly below the diagonal are skipped
//   epilogue: row-sum of squares -> sigma -> mean from the per-slice partials -> LCB / EI -> optional
//   dense stores -> block arg-max carrying (value, lowest index).
// Waves w and w+4 share a SIMD; they take the even and the odd 16-column tiles of the block, so that on the
// diagonal (where tiles below U's diagonal are skipped) both always have nearly the same amount of work and
// keep hiding each other's LDS / branch / DMA-issue latencies.
#include "gpbo_internal.h"

#include <cstdlib>
#include <limits>
#include <mutex>
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 128, BK = 16;  // candidates x columns of V per workgroup, k depth of a tile
// Waves per workgroup and their split.  Measured on MI355X (N = 512): 8 waves as 4 x 2 (wave tile 64 x 64, two
// waves per SIMD) run 0.547 ms per 2^17 candidates; 4 waves with 128 x 64 or 64 x 128 wave tiles (one wave per SIMD,
// 256 accumulator registers) run 1.06 / 0.99 ms: hipcc spills 70-100 VGPRs around the tile loop and one wave
// alone does not keep a SIMD's matrix pipe fed across the LDS reads and the barrier; 16 waves (four per SIMD, wave
// tile 32 x 64 or 64 x 32) run 0.560 / 0.594 ms: half again as many LDS fragment reads per MFMA.
#ifndef GPBO_NW
#define GPBO_NW 8
#endif
#ifndef GPBO_WQ
#define GPBO_WQ 2
#endif
constexpr int NW = GPBO_NW;                // waves per workgroup
constexpr int WQ = GPBO_WQ;                // column groups of waves
constexpr int WR = NW / WQ;                // row groups of waves (BM / WR candidates each)
constexpr int MI = BM / WR / 16;           // 16-row tiles per wave
constexpr int NI = BN / WQ / 16;           // 16-column tiles per wave
static_assert(NI % 2 == 0 && (BM % (WR * 16)) == 0, "tile split");
// (measured on MI355X, N = 512: 256 x 128 runs 0.596 ms per 2^17 candidates, 128 x 256 runs 0.635 ms - the
//  wider column block has more k tiles on the diagonal whose MFMA time is shorter than their 48 KB of loads)
constexpr int LDA = BM + 16;  // padded so lanes l and l+16 (k, k+1) of a ds_read_b64 group hit disjoint banks
constexpr int LDB = BN + 16;
constexpr int A_TILE = BK * LDA;
constexpr int B_TILE = BK * LDB;
constexpr int STAGE = A_TILE + B_TILE;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// 16 bytes per lane, global -> LDS without a VGPR round trip.  The LDS destination is the wave-uniform
// address `l` plus lane*16; the global source is per lane.
__device__ __forceinline__ void glds16(const double *g, double *l) {
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)l, 16, 0, 0);
}

__device__ __forceinline__ bool better(double v2, int64_t i2, double v, int64_t i) { return gpbo_better(v2, i2, v, i); }
__device__ __forceinline__ double acquisition(int kind, double mu, double sigma, double p0, double p1) {
    return gpbo_acquisition(kind, mu, sigma, p0, p1);
}

// GRAM (config 5, q = 8 Monte-Carlo qEI): the tiles of V are computed TRANSPOSED (the two operands of every product swapped:
// register r of lane l then holds V[candidate l & 15][column (l >> 4) + 4r] of its tile), which makes one accumulator
// register at once the A operand (row = candidate, k = column) and the B operand (k = column, column = candidate) of
// v_mfma_f64_16x16x4_f64: four products per finished tile add V V^T of its 16 candidates - the joint posterior's Gram
// blocks of two 8-candidate batches - to 8 more accumulators, and V itself never leaves the registers (round 4 wrote the
// 2.1 GB of V per launch for qei_kernel to read back).  `vbuf` then receives the partial Gram blocks:
// [S * WQ partials][ldk / 8 batches][8 x 8].
template <int VARIANT, bool GRAM = false>  // VARIANT 0 = product; 1, 2 = timing-only diagnostics (GPBO_SIGMA_VARIANT), wrong results
__global__ __launch_bounds__(NW * 64) void sigma_acq_kernel(
    const double *__restrict__ KsT, int64_t ldk, const double *__restrict__ U, int Np,
    const double *__restrict__ mu_part, int nsl, int64_t Mc, double prior_var, int acq_kind, double p0, double p1,
    int64_t idx_base, double *__restrict__ mu_out, double *__restrict__ sigma_out, double *__restrict__ acq_out,
    double *__restrict__ part_val, int64_t *__restrict__ part_idx, unsigned long long *__restrict__ nan_count,
    double *__restrict__ vbuf /* GRAM: the partial Gram blocks of the 8-candidate batches (see above); else unused */,
    double *__restrict__ ss_part /* column-split launches (gridDim.y = S > 1): [S x ldk] partial |v|^2, no epilogue */,
    int xg /* > 1: one-dimensional launch, the xg column groups of a candidate tile 8 linear ids apart (same XCD) */,
    int ntile, int ncb /* > 0: only the first ncb column blocks of V (the prefix-bound screen: |v|^2 over the first
                          128 ncb observations is a LOWER bound of |v|^2, the variance from it an upper bound) */) {
    __shared__ double smem[3 * STAGE];
    // Column split (few candidates, e.g. the re-scoring behind a screen): workgroup (x, s) of S takes the column
    // blocks s, 2S-1-s, 2S+s, 4S-1-s, ... (boustrophedon rounds: block jb costs jb+1 k tiles, so pairing a cheap
    // with an expensive one balances the S workgroups) and leaves its partial row sums for split_finish_kernel.
    int S = (int)gridDim.y, sp = (int)blockIdx.y, tile_x = (int)blockIdx.x;
    if (xg > 1) {
        // Column groups for LARGE launches: with one workgroup per candidate tile, 256 slabs of K*^T (8.4 MB each at
        // N = 4096) are live at a time and every re-read comes from HBM (72 GB per 2^17 candidates, profiles/).  The xg
        // groups of a tile run side by side on one XCD instead (ids 8 apart under the round-robin dispatch - speed only),
        // so 256 / xg slabs are live and the re-reads are served on chip.
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int q = slot / xg;
        S = xg;
        sp = slot - q * xg;
        tile_x = q * 8 + xcd;
        if (tile_x >= ntile) return;
    }
    auto jb_of = [&](int r) { return r * S + ((r & 1) ? (S - 1 - sp) : sp); };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid % WR, wq = wid / WR;  // SIMD partners w, w+4 get different column groups
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t cand0 = (int64_t)tile_x * BM;

    // staging: global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPRs).
    // LDS image rows are padded, and every wave instruction's 1 KiB lies inside one row:
    //   A tile (K*^T): 16 rows x BM/128 pieces, B tile (U): 16 rows x BN/128 pieces; wave w takes pieces w + 8r
    // The tile base pointers are wave-uniform and advance incrementally (scalar adds); each lane's share of a
    // piece is a 32-bit element offset fixed for the whole kernel, so one DMA costs a few scalar instructions.
    const double *a_base = KsT + cand0;
    constexpr int AP = BM / 128, BP = BN / 128;  // 1 KiB pieces per row
    constexpr int NPA = BK * AP / NW, NPB = BK * BP / NW;  // 1-KiB pieces per wave per tile
    unsigned voffA[NPA], voffB[NPB];        // element offsets inside a tile (row * ld + piece * 128 + lane * 2)
    int ldsA[NPA], ldsB[NPB];               // LDS element offsets of the pieces inside a stage
#pragma unroll
    for (int r = 0; r < NPA; ++r) {
        const int u = wid + NW * r, row = u / AP, piece = u % AP;
        voffA[r] = (unsigned)(row * (unsigned)ldk + piece * 128 + lane * 2);
        ldsA[r] = row * LDA + piece * 128;
    }
#pragma unroll
    for (int r = 0; r < NPB; ++r) {
        const int u = wid + NW * r, row = u / BP, piece = u % BP;
        voffB[r] = (unsigned)(row * (unsigned)Np + piece * 128 + lane * 2);
        ldsB[r] = A_TILE + row * LDB + piece * 128;
    }
    d4_t acc[MI][NI];
    d4_t gram[GRAM ? MI : 1];
#pragma unroll
    for (int i = 0; i < (GRAM ? MI : 1); ++i) gram[i] = d4_t{0.0, 0.0, 0.0, 0.0};
    double ss[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 4; ++j) ss[i][j] = 0.0;
    }

    const int nJ = ncb > 0 ? ncb : Np / BN;
    // staging iterator: tile (pj, pk) goes to stage pbuf; pa / pb are its global bases
    int pr = 0, pj = jb_of(0), pk = 0, pbuf = 0;
    const double *pa = a_base, *pb = U + (int64_t)pj * BN;
    auto stage_next = [&]() {
        double *St = smem + pbuf * STAGE;
#pragma unroll
        for (int r = 0; r < NPA; ++r) glds16(pa + voffA[r], St + ldsA[r]);
#pragma unroll
        for (int r = 0; r < NPB; ++r) glds16(pb + voffB[r], St + ldsB[r]);
        pbuf = (pbuf == 2) ? 0 : pbuf + 1;
        if (++pk == (pj + 1) * (BN / BK)) {  // next column block of this workgroup: k restarts
            pj = jb_of(++pr);
            pk = 0;
            pa = a_base;
            pb = U + (int64_t)pj * BN;
        } else {
            pa += (int64_t)BK * ldk;
            pb += (int64_t)BK * Np;
        }
    };

    // Flattened tile sequence (jb, kt): kt = 0 .. 8(jb+1)-1 for jb = 0 .. nJ-1, three LDS stages.
    // The MFMA stream never stops at a tile boundary: the fragments of the next tile's first step are read
    // during the current tile's last step.  The one barrier per tile sits in the MIDDLE of the tile, where
    // every wave still holds fragments for the following MFMAs in registers, so nobody leaves the barrier
    // into an LDS-latency bubble.  At that barrier (iteration t):
    //   - each wave has first waited for its own LDS-DMA of tile t+1 (issued one tile earlier), so after the
    //     barrier tile t+1 is complete for everyone, half a tile before anyone reads it;
    //   - every wave has left tile t-1, so its stage may be overwritten: the DMA of tile t+2 is issued now.
    stage_next();
    if (pj < nJ) stage_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    double a0[MI], b0[NI], a1[MI], b1[NI];
    auto lds_frag = [&](double (&af)[MI], double (&bf)[NI], int buf, int kk) {
        const double *As = smem + buf * STAGE;
        const double *Bs = As + A_TILE;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[mi] = As[(kk + l4) * LDA + wr * (BM / WR) + mi * 16 + l15];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bf[ni] = Bs[(kk + l4) * LDB + (WQ * ni + wq) * 16 + l15];
    };
    lds_frag(a0, b0, 0, 0);
    int dbg_it = 0;
    int cur = 0;
    // One tile.  FULL = every 16x16 tile of U in it is non-zero (all k tiles left of the diagonal block): the
    // MFMA stream is then free of branches; on the diagonal block, MFMA groups of 16-column tiles that lie wholly
    // below U's diagonal are branched over (wave-uniform).  LDS reads, barrier and DMA are the same in both.
    auto tile_body = [&](auto full_tag, int jb, int kt) {
        constexpr bool FULL = decltype(full_tag)::value;
        if (VARIANT == 6 && vbuf && tid == 0) {  // diagnostic: cycle stamp per tile (timing build only)
            vbuf[(int64_t)blockIdx.x * 1024 + dbg_it] = (double)__builtin_amdgcn_s_memtime();
            if (dbg_it == 0 || (jb == nJ - 1 && kt == nJ * (BN / BK) - 1))  // 100 MHz wall clock at both ends (S = 1)
                vbuf[(int64_t)blockIdx.x * 1024 + (dbg_it == 0 ? 1000 : 1001)] = (double)__builtin_amdgcn_s_memrealtime();
            ++dbg_it;
        }
        const int nxt = (cur == 2) ? 0 : cur + 1;
        // first 16-column tile of this wave that still has non-zero rows of U in this k tile (>= 4: none)
        // (wave column group wq owns the 16-column tiles WQ ni + wq of the block, so on the diagonal all waves
        //  keep almost the same amount of work: column tile WQ ni + wq is needed iff it is >= kt')
        int ni_min = (kt - jb * (BN / BK) - wq + WQ - 1) / WQ;
        ni_min = ni_min < 0 ? 0 : ni_min;
        if (VARIANT == 2) ni_min = 0;
        auto mfma8 = [&](const double (&af)[MI], const double (&bf)[NI], int nlo) {
#pragma unroll
            for (int ni = nlo; ni < nlo + NI / 2; ++ni) {
                if (FULL || ni >= ni_min) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[mi][ni] = GRAM ? mfma_f64_16x16x4(bf[ni], af[mi], acc[mi][ni]) : mfma_f64_16x16x4(af[mi], bf[ni], acc[mi][ni]);
                }
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        mfma8(a0, b0, 0);
        lds_frag(a1, b1, cur, 4);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(a0, b0, NI / 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(a1, b1, 0);
        lds_frag(a0, b0, cur, 8);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(a1, b1, NI / 2);
        __builtin_amdgcn_sched_barrier(0);
        if (VARIANT != 1 && VARIANT != 3) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own share of tile t+1 has landed
            __builtin_amdgcn_s_barrier();
        }
        // tile t+2 goes into the stage tile t-1 occupied (free since this barrier).  Issuing the six DMAs keeps a
        // wave from feeding the matrix pipe for a few hundred cycles and the two waves of a SIMD leave the
        // barrier together: column group 0 issues here, column group 1 at the end of the tile.
        const bool do_stage = pj < nJ && VARIANT != 1 && VARIANT != 4;
        if (do_stage && wq == 0) stage_next();
        __builtin_amdgcn_sched_barrier(0);
        mfma8(a0, b0, 0);
        lds_frag(a1, b1, cur, 12);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(a0, b0, NI / 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(a1, b1, 0);
        lds_frag(a0, b0, nxt, 0);  // first step of the next tile (stale but in-bounds after the last tile)
        __builtin_amdgcn_sched_barrier(0);
        mfma8(a1, b1, NI / 2);
        __builtin_amdgcn_sched_barrier(0);
        if (do_stage && wq != 0) stage_next();
        cur = nxt;
    };
    for (int rr = 0, jb = jb_of(0); jb < nJ; jb = jb_of(++rr)) {
        const int heavy = jb * (BN / BK);
        for (int kt = 0; kt < heavy; ++kt) tile_body(std::true_type{}, jb, kt);
        for (int kt = heavy; kt < heavy + BN / BK; ++kt) tile_body(std::false_type{}, jb, kt);
        // column block finished: fold |V|^2 into the row sums (GRAM: V V^T of each tile's 16 candidates into the Gram blocks)
        if (GRAM) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) gram[mi] = mfma_f64_16x16x4(acc[mi][ni][r], acc[mi][ni][r], gram[mi]);
                    acc[mi][ni] = d4_t{0.0, 0.0, 0.0, 0.0};
                }
            continue;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ss[mi][r] = fma(acc[mi][ni][r], acc[mi][ni][r], ss[mi][r]);
                acc[mi][ni] = d4_t{0.0, 0.0, 0.0, 0.0};
            }
    }
    if (GRAM) {
        // this wave's partial Gram of its 64 candidates over its columns: register r of lane l of tile mi is
        // G[(l >> 4) + 4r][l & 15]; the two diagonal 8 x 8 blocks are the batches 2 t and 2 t + 1 of 16-candidate tile t
        const int p = sp * WQ + wq;
        const int64_t nb8 = ldk / 8;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int64_t t16 = (cand0 + wr * (BM / WR) + mi * 16) / 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = l4 + 4 * r;
                const bool lo = (r < 2) && (l15 < 8), hi = (r >= 2) && (l15 >= 8);
                if (lo) vbuf[((int64_t)p * nb8 + 2 * t16) * 64 + i * 8 + l15] = gram[mi][r];
                if (hi) vbuf[((int64_t)p * nb8 + 2 * t16 + 1) * 64 + (i - 8) * 8 + (l15 - 8)] = gram[mi][r];
            }
        }
    }

    // ---- row sums: across the 16 lanes that share a candidate row, then across the two column halves
    __syncthreads();
    double *red = smem;  // [WQ][BM]
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = ss[mi][r];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if (l15 == 0) red[wq * BM + wr * (BM / WR) + mi * 16 + l4 + 4 * r] = v;
        }
    __syncthreads();

    // (all LDS lives in the one smem[] array: a second __shared__ object makes hipcc wait vmcnt(0) for the
    //  LDS-DMA in front of every k tile's first ds_read)
    double *s_val = smem + WQ * BM;
    int64_t *s_idx = reinterpret_cast<int64_t *>(smem + WQ * BM + 4);
    if (ss_part) {  // (kernel-argument uniform) column-split launch: partial sums only
        if (tid < BM) {
            double ssq = red[tid];
#pragma unroll
            for (int q = 1; q < WQ; ++q) ssq += red[q * BM + tid];
            ss_part[(int64_t)sp * ldk + cand0 + tid] = ssq;
        }
        return;
    }
    if (tid < BM) {
        const int64_t c = cand0 + tid;  // chunk-local candidate
        const bool valid = c < Mc;
        double ssq = red[tid];
#pragma unroll
        for (int q = 1; q < WQ; ++q) ssq += red[q * BM + tid];
        double mu = 0.0;
        for (int s = 0; s < nsl; ++s) mu += mu_part[(int64_t)s * ldk + c];
        double var = prior_var - ssq;
        // prefix bound (ncb > 0): the plain pass takes sqrt(|var|), and a variance that rounding has pushed a hair below
        // zero (a candidate on top of an observation) can have a LARGER magnitude than the prefix's: clamp and pad, so that
        // the bound holds whenever the plain variance is above -1e-9 (observed: 1e-13)
        if (ncb > 0) var = fmax(var, 0.0) + GPBO_BOUND_VAR_PAD;
        const double sigma = sqrt(fabs(var));  // abs, then sqrt: point_selector.py:98
        // (prefix bound: the acquisition rounded outward, so that it bounds what the plain pass computes - gpbo_internal.h)
        const double acq = (ncb > 0) ? gpbo_acquisition_ub(acq_kind, mu, sigma, p0, p1) : acquisition(acq_kind, mu, sigma, p0, p1);
        if (valid) {
            if (mu_out) mu_out[c] = mu;
            if (sigma_out) sigma_out[c] = sigma;
            if (acq_out) acq_out[c] = acq;
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

// Final reduction over the per-workgroup partials of all chunks: one workgroup.
__global__ __launch_bounds__(256) void argmax_finish_kernel(const double *__restrict__ part_val,
                                                            const int64_t *__restrict__ part_idx, int64_t nparts,
                                                            const unsigned long long *__restrict__ nan_count,
                                                            gpbo_result *__restrict__ result) {
    __shared__ double s_val[4];
    __shared__ int64_t s_idx[4];
    const int tid = threadIdx.x, lane = tid & 63;
    double bv = -std::numeric_limits<double>::infinity();
    int64_t bi = std::numeric_limits<int64_t>::max();
    for (int64_t p = tid; p < nparts; p += 256)
        if (better(part_val[p], part_idx[p], bv, bi)) { bv = part_val[p]; bi = part_idx[p]; }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double ov = __shfl_xor(bv, off);
        const int64_t oi = __shfl_xor(bi, off);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { s_val[tid >> 6] = bv; s_idx[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (better(s_val[w], s_idx[w], bv, bi)) { bv = s_val[w]; bi = s_idx[w]; }
        result->best_val = bv;
        result->best_idx = bi;
        result->nan_count = (int64_t)*nan_count;
        result->reserved = 0;
    }
}

// Epilogue of a column-split variance launch: |v|^2 = sum of the S partials in index order, then exactly the
// epilogue of sigma_acq_kernel (mean from the per-slice partials, sigma, acquisition, dense stores, block arg-max).
__global__ __launch_bounds__(256) void split_finish_kernel(const double *__restrict__ ss_part, int S, int64_t ldk,
                                                           const double *__restrict__ mu_part, int nsl, int64_t Mc,
                                                           double prior_var, int acq_kind, double p0, double p1,
                                                           int64_t idx_base, double *__restrict__ mu_out,
                                                           double *__restrict__ sigma_out, double *__restrict__ acq_out,
                                                           double *__restrict__ var_out,
                                                           double *__restrict__ part_val, int64_t *__restrict__ part_idx,
                                                           unsigned long long *__restrict__ nan_count,
                                                           double var_pad /* > 0: prefix bound, see sigma_acq_kernel */) {
    __shared__ double s_val[4];
    __shared__ int64_t s_idx[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t c = (int64_t)blockIdx.x * 256 + tid;
    const bool valid = c < Mc;
    double ssq = 0.0, mu = 0.0;
    for (int q = 0; q < S; ++q) ssq += ss_part[(int64_t)q * ldk + c];
    for (int q = 0; q < nsl; ++q) mu += mu_part[(int64_t)q * ldk + c];
    double var = prior_var - ssq;
    if (var_pad > 0.0) var = fmax(var, 0.0) + var_pad;
    const double sigma = sqrt(fabs(var));
    const double acq = (var_pad > 0.0) ? gpbo_acquisition_ub(acq_kind, mu, sigma, p0, p1) : acquisition(acq_kind, mu, sigma, p0, p1);
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
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (better(s_val[w], s_idx[w], bv, bi)) { bv = s_val[w]; bi = s_idx[w]; }
        part_val[blockIdx.x] = bv;
        part_idx[blockIdx.x] = bi;
    }
}

// Acquisition + arg-max over dense mu / sigma that are already on the device (used when the caller asks
// for a second acquisition on the same posterior, e.g. lower_confidence_bound(explore != 4)).
__global__ __launch_bounds__(256) void acq_argmax_kernel(const double *__restrict__ mu, const double *__restrict__ sigma,
                                                         int64_t M, int acq_kind, double p0, double p1, int64_t idx_base,
                                                         double *__restrict__ acq_out, double *__restrict__ part_val,
                                                         int64_t *__restrict__ part_idx,
                                                         unsigned long long *__restrict__ nan_count) {
    __shared__ double s_val[4];
    __shared__ int64_t s_idx[4];
    const int tid = threadIdx.x, lane = tid & 63;
    double bv = -std::numeric_limits<double>::infinity();
    int64_t bi = std::numeric_limits<int64_t>::max();
    unsigned long long nans = 0;
    for (int64_t c = (int64_t)blockIdx.x * 256 + tid; c < M; c += (int64_t)gridDim.x * 256) {
        const double a = acquisition(acq_kind, mu[c], sigma[c], p0, p1);
        if (acq_out) acq_out[c] = a;
        if (a != a) ++nans;
        else if (better(a, idx_base + c, bv, bi)) { bv = a; bi = idx_base + c; }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double ov = __shfl_xor(bv, off);
        const int64_t oi = __shfl_xor(bi, off);
        nans += __shfl_xor(nans, off);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
        s_val[tid >> 6] = bv; s_idx[tid >> 6] = bi;
        if (nans) atomicAdd(nan_count, nans);
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (better(s_val[w], s_idx[w], bv, bi)) { bv = s_val[w]; bi = s_idx[w]; }
        part_val[blockIdx.x] = bv;
        part_idx[blockIdx.x] = bi;
    }
}

// Helper stream + events for the K(X*,X) / variance overlap, one set per device, created on first use and
// kept for the life of the process (nothing is retained about the caller's buffers).  Opt-in only (GPBO_OVERLAP=1, a
// measured null result kept for A/B runs): with it, ONE scoring call per device at a time - two callers on different
// streams would re-record the same events.  The table itself is created under a lock.
struct Helper {
    hipStream_t stream;
    hipEvent_t fork, kdone[2], sdone[2];
};

Helper *helper_for_current_device() {
    static Helper *tab[64] = {nullptr};
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!tab[dev]) {
        Helper *h = new Helper;
        bool ok = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&h->fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 2 && ok; ++i) {
            ok = ok && hipEventCreateWithFlags(&h->kdone[i], hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&h->sdone[i], hipEventDisableTiming) == hipSuccess;
        }
        if (!ok) { delete h; return nullptr; }
        tab[dev] = h;
    }
    return tab[dev];
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct PosteriorLayout {
    int64_t kst_off[2], mup_off[2], xsc_off, prep_off, pval_off, pidx_off, nan_off, ssp_off, total, nparts_cap;
};

// number of column-split workgroups per candidate tile for a launch of nblk tiles (1 = the plain kernel)
int split_factor(int64_t nblk, int64_t nJ, int split_max) {
    if (split_max <= 1 || nblk >= 128) return 1;
    int64_t S = (384 + nblk - 1) / nblk;
    if (S > nJ) S = nJ;
    if (S > split_max) S = split_max;
    return S < 1 ? 1 : (int)S;
}

PosteriorLayout posterior_layout(int64_t Np, int64_t chunk, int64_t M, int split_max = 1) {
    PosteriorLayout L;
    const int64_t nchunks = (M + chunk - 1) / chunk;
    L.nparts_cap = nchunks * (chunk / BM);
    int64_t off = 0;
    // two chunk buffers when there is more than one chunk: K(X*,X) of chunk c+1 is built on a helper
    // stream while the variance kernel of chunk c runs
    const int nbuf = nchunks > 1 ? 2 : 1;
    for (int b = 0; b < 2; ++b) {
        L.kst_off[b] = off; if (b < nbuf) off += align_up((int64_t)sizeof(double) * Np * chunk, 256);
        L.mup_off[b] = off; if (b < nbuf) off += align_up((int64_t)sizeof(double) * (Np / GPBO_KS_SLICE) * chunk, 256);
    }
    if (nbuf == 1) { L.kst_off[1] = L.kst_off[0]; L.mup_off[1] = L.mup_off[0]; }
    L.xsc_off = off; off += align_up((int64_t)sizeof(double) * Np * GPBO_MAX_D, 256);
    L.prep_off = off; off += align_up(gpbo_kstar_mfma_prep_bytes(Np), 256);   // prefix-bound route (kstar_mfma.hip)
    L.pval_off = off; off += align_up((int64_t)sizeof(double) * L.nparts_cap, 256);
    L.pidx_off = off; off += align_up((int64_t)sizeof(int64_t) * L.nparts_cap, 256);
    L.nan_off = off; off += 256;
    L.ssp_off = off;
    off += align_up((int64_t)sizeof(double) * (split_max > 16 ? split_max : 16) * chunk, 256);
    L.total = off;
    return L;
}


// ================================================================================================
// q-point Monte-Carlo Expected Improvement (BASELINE config 5; not in the reference).
// Candidates are grouped consecutively in batches of q = 8.  For a batch b
//     mu_b  = K*_b alpha,      Sigma_b = K_bb - V_b V_b^T,   V_b = rows of V = K* U belonging to the batch,
//     K_bb  = k(x_i, x_j) with the reference's prior diagonal (1 + 1e-4) + 1e-6,
//     qEI_b = 1/S sum_s max(0, max_j (f_best - xi - (mu_b + chol(Sigma_b) z_s)_j)),   z_s fixed base samples.
// V comes from the variance kernel (vbuf); one wave per batch: Gram of 8 rows of V (coalesced row reads),
// wave reduction, 8x8 Cholesky in registers, S samples spread over the lanes, fixed-order reductions.
// ================================================================================================
constexpr int QQ = 8, QT = QQ * (QQ + 1) / 2;  // 36 lower-triangle entries, index i*(i+1)/2 + j, j <= i

struct QeiLs {
    double il2[GPBO_MAX_D];
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void qei_kernel(const double *__restrict__ G, int P, int64_t nb8, const double *__restrict__ mu,
                                                  const double *__restrict__ Xs, int d, QeiLs ls, int64_t nbatch,
                                                  double prior_var, double f_best, double xi,
                                                  const double *__restrict__ Z, int S, int64_t batch_base,
                                                  double *__restrict__ qei_out, double *__restrict__ part_val,
                                                  int64_t *__restrict__ part_idx,
                                                  unsigned long long *__restrict__ nan_count) {
    __shared__ double s_val[4];
    __shared__ int64_t s_idx[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t b = (int64_t)blockIdx.x * 4 + wid;
    const bool valid = b < nbatch;
    const int64_t bb = valid ? b : 0;  // idle waves recompute batch 0 (keeps every shuffle full-wave)
    // V_b V_b^T of the batch: the P partial 8 x 8 blocks the variance launch left (one per column group of its waves and
    // workgroups), summed in partial order - lane e holds entry (e >> 3, e & 7) - then every lane gathers the lower triangle
    double gsum = 0.0;
    for (int q = 0; q < P; ++q) gsum += G[((int64_t)q * nb8 + bb) * 64 + lane];
    double g[QT];
#pragma unroll
    for (int i = 0; i < QQ; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) g[i * (i + 1) / 2 + j] = __shfl(gsum, i * 8 + j);

    // prior covariance of the batch, entry (i, j) by lane t = i(i+1)/2 + j, then gathered by every lane
    double kmine = 0.0;
    if (lane < QT) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= lane) ++i;
        const int j = lane - i * (i + 1) / 2;
        if (i == j) {
            kmine = prior_var;
        } else {
            const double *xi_ = Xs + (bb * QQ + i) * d, *xj_ = Xs + (bb * QQ + j) * d;
            double accd = 0.0;
            for (int k = 0; k < d; ++k) {
                const double diff = xi_[k] - xj_[k];
                accd = fma(diff * diff, ls.il2[k], accd);
            }
            kmine = exp(-0.5 * accd);
        }
    }
    double L[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) L[t] = __shfl(kmine, t) - g[t];  // Sigma_b, lower triangle

    // in-register Cholesky (every lane redundantly; no divergence)
    bool bad = false;
#pragma unroll
    for (int c = 0; c < QQ; ++c) {
        const double piv = L[c * (c + 1) / 2 + c];
        if (!(piv > 0.0)) bad = true;
        const double dd = sqrt(piv), inv = 1.0 / dd;
        L[c * (c + 1) / 2 + c] = dd;
#pragma unroll
        for (int r = c + 1; r < QQ; ++r) L[r * (r + 1) / 2 + c] *= inv;
#pragma unroll
        for (int r = c + 1; r < QQ; ++r)
#pragma unroll
            for (int q2 = c + 1; q2 <= r; ++q2)
                L[r * (r + 1) / 2 + q2] = fma(-L[r * (r + 1) / 2 + c], L[q2 * (q2 + 1) / 2 + c], L[r * (r + 1) / 2 + q2]);
    }
    double m[QQ];
#pragma unroll
    for (int j = 0; j < QQ; ++j) m[j] = f_best - xi - mu[bb * QQ + j];

    double accs = 0.0;
    for (int s = lane; s < S; s += 64) {
        const double *z = Z + (int64_t)s * QQ;
        double zz[QQ];
#pragma unroll
        for (int j = 0; j < QQ; ++j) zz[j] = z[j];
        double best = 0.0;
#pragma unroll
        for (int j = 0; j < QQ; ++j) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k <= j; ++k) t = fma(L[j * (j + 1) / 2 + k], zz[k], t);
            const double imp = m[j] - t;
            best = (imp > best) ? imp : best;  // NaN never raises `best`; it is caught through `bad`
        }
        accs += best;
    }
    double qei = wave_sum(accs) / (double)S;
    bool nanflag = bad || (qei != qei);
#pragma unroll
    for (int j = 0; j < QQ; ++j) nanflag = nanflag || (m[j] != m[j]);
    if (nanflag) qei = __builtin_nan("");
    if (lane == 0) {
        if (valid && qei_out) qei_out[b] = qei;
        if (valid && nanflag) atomicAdd(nan_count, 1ULL);
        s_val[wid] = (valid && !nanflag) ? qei : -std::numeric_limits<double>::infinity();
        s_idx[wid] = (valid && !nanflag) ? batch_base + b : std::numeric_limits<int64_t>::max();
    }
    __syncthreads();
    if (tid == 0) {
        double bv = s_val[0];
        int64_t bi = s_idx[0];
        for (int w = 1; w < 4; ++w)
            if (better(s_val[w], s_idx[w], bv, bi)) { bv = s_val[w]; bi = s_idx[w]; }
        part_val[blockIdx.x] = bv;
        part_idx[blockIdx.x] = bi;
    }
}

struct QeiLayout {
    int64_t kst_off, mup_off, xsc_off, v_off, mu_off, spv_off, spi_off, pval_off, pidx_off, nan_off, ssp_off, total;
};

QeiLayout qei_layout(int64_t Np, int64_t chunk, int64_t M) {
    QeiLayout L;
    const int64_t nchunks = (M + chunk - 1) / chunk;
    int64_t off = 0;
    L.kst_off = off; off += align_up((int64_t)sizeof(double) * Np * chunk, 256);
    L.mup_off = off; off += align_up((int64_t)sizeof(double) * (Np / GPBO_KS_SLICE) * chunk, 256);
    L.xsc_off = off; off += align_up((int64_t)sizeof(double) * Np * GPBO_MAX_D, 256);
    L.v_off = off; off += align_up((int64_t)sizeof(double) * 64 * (chunk / 8) * 16 * WQ, 256);   // Gram blocks: <= 16 x WQ partials
    L.mu_off = off; off += align_up((int64_t)sizeof(double) * chunk, 256);
    L.spv_off = off; off += align_up((int64_t)sizeof(double) * (chunk / BM), 256);
    L.spi_off = off; off += align_up((int64_t)sizeof(int64_t) * (chunk / BM), 256);
    const int64_t nparts = nchunks * ((chunk / QQ + 3) / 4);
    L.pval_off = off; off += align_up((int64_t)sizeof(double) * nparts, 256);
    L.pidx_off = off; off += align_up((int64_t)sizeof(int64_t) * nparts, 256);
    L.nan_off = off; off += 256;
    L.ssp_off = off; off += align_up((int64_t)sizeof(double) * 16 * chunk, 256);  // column-group partials (large calls)
    L.total = off;
    return L;
}

}  // namespace

int gpbo_launch_split_finish(const double *ss_part, int S, int64_t ldk, const double *mu_part, int nsl, int64_t Mc,
                             double prior_var, int acq_kind, double p0, double p1, int64_t idx_base, double *mu_out,
                             double *sigma_out, double *acq_out, double *var_out, double *part_val, int64_t *part_idx,
                             unsigned long long *nan_count, hipStream_t st) {
    const int64_t nb = (Mc + 255) / 256;
    hipLaunchKernelGGL(split_finish_kernel, dim3((unsigned)nb), dim3(256), 0, st, ss_part, S, ldk, mu_part, nsl, Mc, prior_var,
                       acq_kind, p0, p1, idx_base, mu_out, sigma_out, acq_out, var_out, part_val, part_idx, nan_count, 0.0);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

int gpbo_launch_argmax_finish(const double *part_val, const int64_t *part_idx, int64_t nparts,
                              const unsigned long long *nan_count, gpbo_result *result, hipStream_t st) {
    hipLaunchKernelGGL(argmax_finish_kernel, dim3(1), dim3(256), 0, st, part_val, part_idx, nparts, nan_count, result);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

extern "C" int64_t gpbo_posterior_workspace_bytes(int64_t Np, int64_t chunk, int64_t M) {
    if (Np < GPBO_NPAD || Np % GPBO_NPAD || chunk < GPBO_CHUNK_GRANULE || chunk % GPBO_CHUNK_GRANULE || chunk > GPBO_CHUNK_MAX || M < 1)
        return GPBO_ERR_ARG;
    return posterior_layout(Np, chunk, M).total;
}

extern "C" int gpbo_posterior_acq_f64(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                                      const double *ls_host, const double *U, const double *alpha, double prior_var,
                                      int32_t acq_kind, double p0, double p1, double diag_add, int64_t idx_offset,
                                      int64_t chunk, double *mu_out, double *sigma_out, double *acq_out,
                                      gpbo_result *result, void *work, int64_t work_bytes, gpbo_profile *prof,
                                      void *stream) {
    return gpbo_posterior_acq_f64_split(Xs, M, X, N, Np, d, ls_host, U, alpha, prior_var, acq_kind, p0, p1, diag_add,
                                        idx_offset, chunk, mu_out, sigma_out, acq_out, result, work, work_bytes, prof, 1,
                                        0, stream);
}

// The prefix-bound screen's first pass (rescore.hip, gpbo_bound_select_f64): the mean over all N observations, the
// variance product over the FIRST n_prefix columns of V only.  |v_c|^2 summed over a prefix of its components is a lower
// bound of the whole sum, so sigma_out / acq_out hold UPPER bounds of the fp64 path's values (both acquisitions increase
// with sigma; LCB only for explore >= 0).  K*^T is stored for the first n_prefix observations only.
extern "C" int gpbo_posterior_prefix_f64(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                                         const double *ls_host, const double *U, const double *alpha, double prior_var,
                                         int32_t acq_kind, double p0, double p1, int64_t idx_offset, int64_t chunk,
                                         int64_t n_prefix, double *mu_out, double *sigma_ub_out, double *acq_ub_out,
                                         gpbo_result *result, void *work, int64_t work_bytes, gpbo_profile *prof,
                                         void *stream) {
    if (n_prefix < BN || n_prefix % BN || n_prefix > Np) return GPBO_ERR_ARG;
    if (acq_kind == GPBO_ACQ_LCB && !(p0 >= 0.0)) return GPBO_ERR_ARG;
    return gpbo_posterior_acq_f64_split(Xs, M, X, N, Np, d, ls_host, U, alpha, prior_var, acq_kind, p0, p1, 0.0,
                                        idx_offset, chunk, mu_out, sigma_ub_out, acq_ub_out, result, work, work_bytes,
                                        prof, 1, n_prefix, stream);
}

int64_t gpbo_posterior_workspace_bytes_split(int64_t Np, int64_t chunk, int64_t M, int split_max) {
    if (Np < GPBO_NPAD || Np % GPBO_NPAD || chunk < GPBO_CHUNK_GRANULE || chunk % GPBO_CHUNK_GRANULE || chunk > GPBO_CHUNK_MAX || M < 1)
        return GPBO_ERR_ARG;
    return posterior_layout(Np, chunk, M, split_max).total;
}

// split_max > 1: launches of fewer than 128 candidate tiles are split over the column blocks of V (the re-scoring
// of a few survivors: one workgroup per 256 candidates would take N^2/2 MFMA-bound k tiles on ONE compute unit).
// The partial sums are combined in a fixed order, so results are deterministic; they differ from the unsplit
// kernel's by the rounding of a different summation order (~1e-16 relative).
int gpbo_posterior_acq_f64_split(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                                 const double *ls_host, const double *U, const double *alpha, double prior_var,
                                 int32_t acq_kind, double p0, double p1, double diag_add, int64_t idx_offset,
                                 int64_t chunk, double *mu_out, double *sigma_out, double *acq_out, gpbo_result *result,
                                 void *work, int64_t work_bytes, gpbo_profile *prof, int split_max,
                                 int64_t n_prefix /* 0: everything; else see gpbo_posterior_prefix_f64 */, void *stream) {
    if (!Xs || !X || !U || !alpha || !result || !work) return GPBO_ERR_ARG;
    if (n_prefix < 0 || n_prefix > Np || n_prefix % BN || (n_prefix && diag_add != 0.0)) return GPBO_ERR_ARG;
    if (M < 1 || N < 1 || Np != gpbo_padded_n(N) || Np > (1 << 20)) return GPBO_ERR_ARG;
    if (chunk < GPBO_CHUNK_GRANULE || chunk % GPBO_CHUNK_GRANULE || chunk > GPBO_CHUNK_MAX) return GPBO_ERR_ARG;
    if (acq_kind != GPBO_ACQ_LCB && acq_kind != GPBO_ACQ_EI) return GPBO_ERR_ARG;
    if (((uintptr_t)work & 255) || ((uintptr_t)U & 15)) return GPBO_ERR_ARG;
    const PosteriorLayout L = posterior_layout(Np, chunk, M, split_max);
    if (work_bytes < L.total) return GPBO_ERR_WORKSPACE;
    hipStream_t st = gpbo_stream(stream);
    char *w = reinterpret_cast<char *>(work);
    double *ss_part = reinterpret_cast<double *>(w + L.ssp_off);
    double *KsT[2] = {reinterpret_cast<double *>(w + L.kst_off[0]), reinterpret_cast<double *>(w + L.kst_off[1])};
    double *mu_part[2] = {reinterpret_cast<double *>(w + L.mup_off[0]), reinterpret_cast<double *>(w + L.mup_off[1])};
    double *Xsc = reinterpret_cast<double *>(w + L.xsc_off);
    double *part_val = reinterpret_cast<double *>(w + L.pval_off);
    int64_t *part_idx = reinterpret_cast<int64_t *>(w + L.pidx_off);
    unsigned long long *nan_count = reinterpret_cast<unsigned long long *>(w + L.nan_off);
    // prefix-bound route: K(X*,X) with the pair distances on the matrix cores and the mean reported from below by its error
    // bound (kstar_mfma.hip; GPBO_PREFIX_VALU=1: the difference-form kernel instead, for A/B runs)
    static const bool prefix_valu = getenv("GPBO_PREFIX_VALU") && atoi(getenv("GPBO_PREFIX_VALU"));
    static const bool overlap_env = getenv("GPBO_OVERLAP") && atoi(getenv("GPBO_OVERLAP"));
    // (and only while the unused rows of the K*^T slab can take that launch's mean partials)
    const bool kstar_mfma = n_prefix > 0 && !prefix_valu && !overlap_env && n_prefix + n_prefix / GPBO_KS_SLICE <= Np;
    const bool anyd = d > GPBO_MAX_D;   // slow path of the fp64 route: no unrolled registers, no pre-scaled copy
    if (anyd && (n_prefix > 0 || d > GPBO_MAX_D_ANY)) return GPBO_ERR_ARG;
    if (anyd) {
        if (hipMemsetAsync(nan_count, 0, sizeof(unsigned long long), st) != hipSuccess) return GPBO_ERR_LAUNCH;
    } else {
        // observations / (ls sqrt 2), once per call; the same launch clears the NaN counter
        int rc0 = gpbo_scale_points_launch(X, N, Np, d, ls_host, Xsc, nan_count, stream);
        if (rc0 != GPBO_OK) return rc0;
    }
    void *prep_buf = w + L.prep_off;
    if (kstar_mfma) {
        int rc0 = gpbo_kstar_mfma_prep(X, N, Np, d, ls_host, alpha, prep_buf, stream);
        if (rc0 != GPBO_OK) return rc0;
    }

    // Timing-only variants of the variance kernel (wrong results; tools/tile_stamps.py uses 6) exist only in a
    // diagnostics build (GPBO_DIAG=1 build.sh): the shipped library has no switch into them.
#ifdef GPBO_DIAGNOSTICS
    static const int variant = getenv("GPBO_SIGMA_VARIANT") ? atoi(getenv("GPBO_SIGMA_VARIANT")) : 0;
#else
    constexpr int variant = 0;
#endif
    // Measured on MI355X (N=512, M=2^20): running K(X*,X) of chunk c+1 beside the variance kernel of chunk c gains
    // nothing - the variance launches slow down by what the overlap hides (0.59 -> 0.70 ms), i.e. fp64 VALU work
    // and fp64 MFMA work do not co-execute on gfx950.  Kept as an opt-in (GPBO_OVERLAP=1) for other shapes.
    const int64_t nchunks = (M + chunk - 1) / chunk;
    Helper *hp = (nchunks > 1 && overlap_env) ? helper_for_current_device() : nullptr;
    // Fork: the helper stream builds K(X*,X)+mu of chunk c+1 (fp64 VALU + HBM writes) while the caller's
    // stream runs the variance kernel of chunk c (matrix cores); two chunk buffers, events both ways.
    hipStream_t ks = hp ? hp->stream : st;
    if (hp) {
        if (hipEventRecord(hp->fork, st) != hipSuccess || hipStreamWaitEvent(ks, hp->fork, 0) != hipSuccess)
            return GPBO_ERR_LAUNCH;
    }
    // partials of the mean per candidate: 64-observation slices, coarser ones from the matrix-core K(X*,X) kernel
    const int nsl = kstar_mfma ? (int)(Np / gpbo_kstar_mfma_slice(Np)) : (int)(Np / GPBO_KS_SLICE);
    bool prev_recorded = false;  // the previous chunk's variance launch has an end event in the slot before
    auto launch_kstar = [&](int64_t c) -> int {
        const int64_t s = c * chunk;
        const int64_t Mc = (M - s < chunk) ? (M - s) : chunk;
        const int b = (int)(c & 1);
        // K(X*,X) launches are timed only in the plain in-order mode, where the launches of this call form one chain
        // on the caller's stream: the event in front of the variance launch of the slot ends the K(X*,X) interval, and
        // the interval begins at the previous variance launch's end event (chunks after the first) or at kbegin.
        const bool krec = prof && !hp && prof->count < prof->capacity;
        if (krec) {
            const bool chained = c > 0 && prof->count > 0 && prev_recorded;
            prof->kmode[prof->count] = chained ? 2 : 1;
            if (!chained &&
                hipEventRecord(reinterpret_cast<hipEvent_t>(prof->kbegin[prof->count]), ks) != hipSuccess)
                return GPBO_ERR_LAUNCH;
        } else if (prof && prof->count < prof->capacity) {
            prof->kmode[prof->count] = 0;
        }
        int rc;
        if (kstar_mfma) {
            // The MEAN of all N observations from the matrix-core kernel (expanded distances; reported from below by its
            // own error bound), but the n_prefix stored rows of K*^T from the difference-form kernel of the plain pass:
            // the expanded form's entry error grows with the points' distance from the centroid in length-scale units
            // (unnormalised inputs), is amplified by |U| and would need a data-dependent variance pad; with the plain
            // pass's own entries |v[:J]|^2 is a partial sum of the very squares the plain pass adds up, so the bound
            // holds for any inputs.  Costs n_prefix / N of the fp64-VALU kernel: 6 % at N / 16.
            rc = gpbo_kstar_mu_mfma(Xs + s * d, Mc, N, Np, d, ls_host, alpha, prep_buf, KsT[b], chunk, mu_part[b], 0, ks);
            if (rc != GPBO_OK) return rc;
            const int64_t nrow = (N < n_prefix) ? N : n_prefix;
            // (its mean partials - of the first n_prefix observations only - are not wanted: they go to rows of the K*^T
            //  slab that the prefix mode neither writes nor reads, [n_prefix, n_prefix + n_prefix / 64))
            rc = gpbo_kstar_mu_rows(Xs + s * d, Mc, Xsc, nrow, n_prefix, d, ls_host, alpha, 0.0, idx_offset + s, KsT[b], chunk,
                                    KsT[b] + n_prefix * chunk, n_prefix, ks);
        } else if (anyd) {
            rc = gpbo_kstar_mu_anyd(Xs + s * d, Mc, X, N, Np, d, ls_host, alpha, diag_add, idx_offset + s, KsT[b], chunk,
                                    mu_part[b], ks);
        } else {
            rc = gpbo_kstar_mu_rows(Xs + s * d, Mc, Xsc, N, Np, d, ls_host, alpha, diag_add, idx_offset + s, KsT[b], chunk,
                                    mu_part[b], n_prefix ? n_prefix : Np, ks);
        }
        if (rc != GPBO_OK) return rc;
        if (hp && hipEventRecord(hp->kdone[b], ks) != hipSuccess) return GPBO_ERR_LAUNCH;
        return GPBO_OK;
    };
    int rc = launch_kstar(0);
    if (rc != GPBO_OK) return rc;
    int64_t nparts = 0;
    for (int64_t c = 0; c < nchunks; ++c) {
        const int64_t s = c * chunk;
        const int64_t Mc = (M - s < chunk) ? (M - s) : chunk;
        const int b = (int)(c & 1);
        if (hp && c + 1 < nchunks) {
            // buffer (c+1)&1 was last read by the variance kernel of chunk c-1
            if (c >= 1 && hipStreamWaitEvent(ks, hp->sdone[(c + 1) & 1], 0) != hipSuccess) return GPBO_ERR_LAUNCH;
            rc = launch_kstar(c + 1);
            if (rc != GPBO_OK) return rc;
        }
        if (hp && hipStreamWaitEvent(st, hp->kdone[b], 0) != hipSuccess) return GPBO_ERR_LAUNCH;
        const int64_t nblk = (Mc + BM - 1) / BM;
        const bool rec = prof && prof->count < prof->capacity;
        if (rec && hipEventRecord(reinterpret_cast<hipEvent_t>(prof->begin[prof->count]), st) != hipSuccess)
            return GPBO_ERR_LAUNCH;
        const int S = split_factor(nblk, (n_prefix ? n_prefix : Np) / BN, split_max);
        // Column groups on one XCD for large calls (see the kernel): measured on MI355X at N = 4096, 2^21 candidates,
        // same box: 543 -> 509 ms per step with 8 groups (16: 512), the variance launches 32.9 -> 30.7 ms.  The rule depends
        // on the problem (N, candidates of the CALL), never on the chunking, so results stay chunk-size invariant bit for
        // bit.  GPBO_F64_GROUPS=1 switches it off (A/B runs).
        static const int xg_env = getenv("GPBO_F64_GROUPS") ? atoi(getenv("GPBO_F64_GROUPS")) : 8;
        if (xg_env > 1 && S == 1 && M >= 32768 && Np / BN >= 2 * xg_env && xg_env <= 16 && n_prefix == 0) {
            const int64_t grid1 = (nblk + 7) / 8 * 8 * xg_env;
            hipLaunchKernelGGL(sigma_acq_kernel<0>, dim3((unsigned)grid1), dim3(NW * 64), 0, st, KsT[b], chunk, U, (int)Np,
                               mu_part[b], nsl, Mc, prior_var, (int)acq_kind, p0, p1, idx_offset + s,
                               (double *)nullptr, (double *)nullptr, (double *)nullptr, part_val + nparts,
                               part_idx + nparts, nan_count, (double *)nullptr, ss_part, xg_env, (int)nblk, 0);
            hipLaunchKernelGGL(split_finish_kernel, dim3((unsigned)nblk), dim3(256), 0, st, ss_part, xg_env, chunk, mu_part[b],
                               nsl, Mc, prior_var, (int)acq_kind, p0, p1, idx_offset + s,
                               mu_out ? mu_out + s : nullptr, sigma_out ? sigma_out + s : nullptr,
                               acq_out ? acq_out + s : nullptr, (double *)nullptr, part_val + nparts, part_idx + nparts,
                               nan_count, n_prefix ? GPBO_BOUND_VAR_PAD : 0.0);
        } else if (S > 1) {
            hipLaunchKernelGGL(sigma_acq_kernel<0>, dim3((unsigned)nblk, (unsigned)S), dim3(NW * 64), 0, st, KsT[b], chunk, U,
                               (int)Np, mu_part[b], nsl, Mc, prior_var, (int)acq_kind, p0, p1,
                               idx_offset + s, (double *)nullptr, (double *)nullptr, (double *)nullptr, part_val + nparts,
                               part_idx + nparts, nan_count, (double *)nullptr, ss_part, 1, (int)nblk, (int)(n_prefix / BN));
            hipLaunchKernelGGL(split_finish_kernel, dim3((unsigned)nblk), dim3(256), 0, st, ss_part, S, chunk, mu_part[b],
                               nsl, Mc, prior_var, (int)acq_kind, p0, p1, idx_offset + s,
                               mu_out ? mu_out + s : nullptr, sigma_out ? sigma_out + s : nullptr,
                               acq_out ? acq_out + s : nullptr, (double *)nullptr, part_val + nparts, part_idx + nparts,
                               nan_count, n_prefix ? GPBO_BOUND_VAR_PAD : 0.0);
        } else {
#define GPBO_SIGMA_LAUNCH(V)                                                                                        \
    hipLaunchKernelGGL(sigma_acq_kernel<V>, dim3((unsigned)nblk), dim3(NW * 64), 0, st, KsT[b], chunk, U, (int)Np,          \
                       mu_part[b], nsl, Mc, prior_var, (int)acq_kind, p0, p1, idx_offset + s,               \
                       mu_out ? mu_out + s : nullptr, sigma_out ? sigma_out + s : nullptr,                              \
                       acq_out ? acq_out + s : nullptr, part_val + nparts, part_idx + nparts, nan_count,                     \
                       (V == 6 && c == nchunks - 1 && nchunks > 1) ? KsT[(c + 1) & 1] : (double *)nullptr, (double *)nullptr, 1, \
                       (int)nblk, (int)(n_prefix / BN))
#ifdef GPBO_DIAGNOSTICS
        if (variant == 1) GPBO_SIGMA_LAUNCH(1);
        else if (variant == 2) GPBO_SIGMA_LAUNCH(2);
        else if (variant == 3) GPBO_SIGMA_LAUNCH(3);
        else if (variant == 4) GPBO_SIGMA_LAUNCH(4);
        else if (variant == 6) GPBO_SIGMA_LAUNCH(6);
        else GPBO_SIGMA_LAUNCH(0);
#else
        (void)variant;
        GPBO_SIGMA_LAUNCH(0);
#endif
#undef GPBO_SIGMA_LAUNCH
        }
        if (rec) {
            if (hipEventRecord(reinterpret_cast<hipEvent_t>(prof->end[prof->count]), st) != hipSuccess)
                return GPBO_ERR_LAUNCH;
            prof->cands[prof->count] = Mc;
            ++prof->count;
        }
        prev_recorded = rec;
        GPBO_CHECK_LAUNCH();
        if (hp && hipEventRecord(hp->sdone[b], st) != hipSuccess) return GPBO_ERR_LAUNCH;
        if (!hp && c + 1 < nchunks) {
            rc = launch_kstar(c + 1);
            if (rc != GPBO_OK) return rc;
        }
        nparts += nblk;
    }
    hipLaunchKernelGGL(argmax_finish_kernel, dim3(1), dim3(256), 0, st, part_val, part_idx, nparts, nan_count, result);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

extern "C" int64_t gpbo_acq_workspace_bytes(void) { return 1024 * 16 + 256; }

extern "C" int gpbo_acq_argmax_f64(const double *mu, const double *sigma, int64_t M, int32_t acq_kind, double p0,
                                   double p1, int64_t idx_offset, double *acq_out, gpbo_result *result, void *work,
                                   int64_t work_bytes, void *stream) {
    if (!mu || !sigma || !result || !work || M < 1) return GPBO_ERR_ARG;
    if (acq_kind != GPBO_ACQ_LCB && acq_kind != GPBO_ACQ_EI) return GPBO_ERR_ARG;
    if (work_bytes < gpbo_acq_workspace_bytes()) return GPBO_ERR_WORKSPACE;
    hipStream_t st = gpbo_stream(stream);
    char *w = reinterpret_cast<char *>(work);
    double *part_val = reinterpret_cast<double *>(w);
    int64_t *part_idx = reinterpret_cast<int64_t *>(w + 1024 * 8);
    unsigned long long *nan_count = reinterpret_cast<unsigned long long *>(w + 1024 * 16);
    if (hipMemsetAsync(nan_count, 0, sizeof(unsigned long long), st) != hipSuccess) return GPBO_ERR_LAUNCH;
    int64_t nblk = (M + 255) / 256;
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(acq_argmax_kernel, dim3((unsigned)nblk), dim3(256), 0, st, mu, sigma, M, (int)acq_kind, p0, p1,
                       idx_offset, acq_out, part_val, part_idx, nan_count);
    hipLaunchKernelGGL(argmax_finish_kernel, dim3(1), dim3(256), 0, st, part_val, part_idx, nblk, nan_count, result);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

extern "C" int64_t gpbo_qei_workspace_bytes(int64_t Np, int64_t chunk, int64_t M) {
    if (Np < GPBO_NPAD || Np % GPBO_NPAD || chunk < GPBO_CHUNK_GRANULE || chunk % GPBO_CHUNK_GRANULE || chunk > GPBO_CHUNK_MAX || M < 1)
        return GPBO_ERR_ARG;
    return qei_layout(Np, chunk, M).total;
}

extern "C" int gpbo_posterior_qei_f64(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                                      const double *ls_host, const double *U, const double *alpha, double prior_var,
                                      double f_best, double xi, const double *Z, int32_t S, int64_t batch_offset,
                                      int64_t chunk, double *qei_out, gpbo_result *result, void *work,
                                      int64_t work_bytes, gpbo_profile *prof, void *stream) {
    if (!Xs || !X || !U || !alpha || !Z || !result || !work) return GPBO_ERR_ARG;
    if (M < QQ || M % QQ || N < 1 || Np != gpbo_padded_n(N) || S < 1 || d < 1 || d > GPBO_MAX_D) return GPBO_ERR_ARG;
    if (chunk < GPBO_CHUNK_GRANULE || chunk % GPBO_CHUNK_GRANULE || chunk > GPBO_CHUNK_MAX) return GPBO_ERR_ARG;
    if (((uintptr_t)work & 255) || ((uintptr_t)U & 15)) return GPBO_ERR_ARG;
    const QeiLayout L = qei_layout(Np, chunk, M);
    if (work_bytes < L.total) return GPBO_ERR_WORKSPACE;
    QeiLs ls;
    for (int k = 0; k < GPBO_MAX_D; ++k) ls.il2[k] = 0.0;
    for (int k = 0; k < d; ++k) {
        if (!(ls_host[k] > 0.0)) return GPBO_ERR_ARG;
        ls.il2[k] = 1.0 / (ls_host[k] * ls_host[k]);
    }
    hipStream_t st = gpbo_stream(stream);
    char *w = reinterpret_cast<char *>(work);
    double *KsT = reinterpret_cast<double *>(w + L.kst_off);
    double *mu_part = reinterpret_cast<double *>(w + L.mup_off);
    double *Xsc = reinterpret_cast<double *>(w + L.xsc_off);
    double *Vb = reinterpret_cast<double *>(w + L.v_off);
    double *mu = reinterpret_cast<double *>(w + L.mu_off);
    double *spv = reinterpret_cast<double *>(w + L.spv_off);
    int64_t *spi = reinterpret_cast<int64_t *>(w + L.spi_off);
    double *part_val = reinterpret_cast<double *>(w + L.pval_off);
    int64_t *part_idx = reinterpret_cast<int64_t *>(w + L.pidx_off);
    unsigned long long *nan_count = reinterpret_cast<unsigned long long *>(w + L.nan_off);
    unsigned long long *nan_scratch = nan_count + 8;  // NaN count of the single-point pass (not reported)
    if (hipMemsetAsync(nan_count, 0, 256, st) != hipSuccess) return GPBO_ERR_LAUNCH;
    int rc = gpbo_scale_points_f64(X, N, Np, d, ls_host, Xsc, stream);
    if (rc != GPBO_OK) return rc;
    int64_t nparts = 0;
    for (int64_t s = 0; s < M; s += chunk) {
        const int64_t Mc = (M - s < chunk) ? (M - s) : chunk;
        // one profile slot per chunk: kbegin | K(X*,X) | begin | variance launch (+ its finish) | end | qEI launch | qend
        const bool rec = prof && prof->count < prof->capacity;
        auto mark = [&](void *ev) { return hipEventRecord(reinterpret_cast<hipEvent_t>(ev), st) == hipSuccess; };
        if (rec && !mark(prof->kbegin[prof->count])) return GPBO_ERR_LAUNCH;
        rc = gpbo_kstar_mu_f64(Xs + s * d, Mc, Xsc, N, Np, d, ls_host, alpha, 0.0, 0, KsT, chunk, mu_part, stream);
        if (rc != GPBO_OK) return rc;
        if (rec && !mark(prof->begin[prof->count])) return GPBO_ERR_LAUNCH;
        const int64_t nblk = (Mc + BM - 1) / BM;
        // the variance kernel in its GRAM form leaves the batches' partial V V^T blocks (vbuf) and mu; its own single-point
        // acquisition result is ignored.  Large calls: column groups on one XCD as in gpbo_posterior_acq_f64 (each group
        // leaves the Gram partials of its column blocks; the mean then comes from split_finish_kernel).
        static const int xg_env = getenv("GPBO_F64_GROUPS") ? atoi(getenv("GPBO_F64_GROUPS")) : 8;
        int gram_parts = WQ;   // partial Gram blocks per batch: one per column group of waves and of workgroups
        if (xg_env > 1 && xg_env <= 16 && M >= 32768 && Np / BN >= 2 * xg_env) {
            gram_parts = xg_env * WQ;
            double *ss_part = reinterpret_cast<double *>(w + L.ssp_off);
            const int64_t grid1 = (nblk + 7) / 8 * 8 * xg_env;
            hipLaunchKernelGGL((sigma_acq_kernel<0, true>), dim3((unsigned)grid1), dim3(NW * 64), 0, st, KsT, chunk, U, (int)Np, mu_part,
                               (int)(Np / GPBO_KS_SLICE), Mc, prior_var, (int)GPBO_ACQ_LCB, 0.0, 0.0, (int64_t)0,
                               (double *)nullptr, (double *)nullptr, (double *)nullptr, spv, spi, nan_scratch, Vb, ss_part,
                               xg_env, (int)nblk, 0);
            hipLaunchKernelGGL(split_finish_kernel, dim3((unsigned)nblk), dim3(256), 0, st, ss_part, xg_env, chunk, mu_part,
                               (int)(Np / GPBO_KS_SLICE), Mc, prior_var, (int)GPBO_ACQ_LCB, 0.0, 0.0, (int64_t)0, mu,
                               (double *)nullptr, (double *)nullptr, (double *)nullptr, spv, spi, nan_scratch, 0.0);
        } else {
            hipLaunchKernelGGL((sigma_acq_kernel<0, true>), dim3((unsigned)nblk), dim3(NW * 64), 0, st, KsT, chunk, U, (int)Np, mu_part,
                               (int)(Np / GPBO_KS_SLICE), Mc, prior_var, (int)GPBO_ACQ_LCB, 0.0, 0.0, (int64_t)0, mu,
                               (double *)nullptr, (double *)nullptr, spv, spi, nan_scratch, Vb, (double *)nullptr, 1,
                               (int)nblk, 0);
        }
        GPBO_CHECK_LAUNCH();
        if (rec && !mark(prof->end[prof->count])) return GPBO_ERR_LAUNCH;
        const int64_t nbatch = Mc / QQ;
        const int64_t qblk = (nbatch + 3) / 4;
        hipLaunchKernelGGL(qei_kernel, dim3((unsigned)qblk), dim3(256), 0, st, Vb, gram_parts, chunk / 8, mu, Xs + s * d, (int)d, ls, nbatch,
                           prior_var, f_best, xi, Z, (int)S, batch_offset + s / QQ, qei_out ? qei_out + s / QQ : nullptr,
                           part_val + nparts, part_idx + nparts, nan_count);
        GPBO_CHECK_LAUNCH();
        if (rec) {
            if (!mark(prof->qend[prof->count])) return GPBO_ERR_LAUNCH;
            prof->kmode[prof->count] = 1;
            prof->qmode[prof->count] = 1;
            prof->cands[prof->count] = Mc;
            ++prof->count;
        }
        nparts += qblk;
    }
    return gpbo_launch_argmax_finish(part_val, part_idx, nparts, nan_count, result, st);
}
