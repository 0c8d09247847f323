// Variance product on the INTEGER matrix cores: V = K* U from int8 slices (Ozaki-style splitting), exact integer
// accumulation, fp64 recombination.  Same mathematics as sigma_acq.hip (reference: point_selector.py:91,98):
//     sigma_c^2 = prior_var - |U^T k_c|^2,
// but the N^2 multiply-adds per candidate run on v_mfma_i32_32x32x32_i8 (64x the fp64 MFMA rate per clock) instead of
// v_mfma_f64_16x16x4_f64.
//
// Splitting.  k in [0, 1] and the column-scaled U_ij 2^-e_j in [-1, 1] are rounded ONCE to fixed point,
//     T_k = rint(k 2^38)  (five digits),      T_u = rint(U_ij 2^-e_j 2^46)  (six digits),
// and written in balanced base-256 digits  T = sum_a D_a 256^(n-1-a),  D_a in [-128, 127]  (int8 slices; the digits come
// out of T + 0x80..80 byte by byte, xor 0x80).  Digit a of either operand has weight 2^(-6-8a), so
//     v_j = 2^e_j sum_{a,b} 2^(-12 - 8(a+b)) (K_a U_b)_j
// where every K_a U_b is an EXACT integer (|sum| <= N 2^14 < 2^31 for N <= 16384: int32 accumulators).  The digit
// pairs are accumulated per diagonal g = a + b (six int32 accumulator sets) and pairs with a + b > 5 are dropped:
// 20 slice products (a sixth digit of K* would only meet U's first: pair (5,0), below the other dropped terms).
// Error (tools/ozaki_error.py, the benchmark problem, "sk=5 su=6 keep=6"): the dropped diagonals and the two roundings
// leave |dv_j| <= 2.7e-10 and |dsigma| <= 1.7e-10 at N = 4096 (fp64 MFMA path: 1.3e-13); with five digits of U as well
// (19 products) it would be 1.8e-9, with 15 products 2e-8.  Everything after the two roundings is exact integer
// arithmetic, so the result does not depend on tile shapes, chunking or summation order.
// The arg-max is still decided in fp64 (rescore.hip) - this pass is a screen with a 1e-10-accurate variance.
//
// Geometry.  Operands live in HBM as ready-made MFMA fragments: 1-KiB blocks [32 rows x 32 k] in lane order (lane l =
// row l & 31, k half l >> 5, 16 consecutive k per lane), indexed [k block][row tile][slice].  A stage (32 k) of a
// 128 x 128 block tile is two contiguous pieces (4 row tiles x 5 slices of K* = 20 KiB, 4 column tiles x 6 slices of
// U = 24 KiB): 44 LDS-DMA instructions, fragments land in LDS exactly as ds_read_b128 will fetch them (no bank
// conflicts, no repacking).  Workgroup = 512 threads = 8 waves as 2 (candidates) x 4 (columns): wave tile 64 x 32 = two
// 32 x 32 MFMA tiles x 6 diagonals = 192 accumulator registers; 40 MFMAs per wave and stage; three-stage LDS ring
// (132 KiB), one barrier per stage.
//
// What bounds it (MI355X, N = 4096, PMC passes in profiles/): the matrix pipes are 68 % busy; the rest is waiting for
// operands - 44 KiB per stage and compute unit is 12 B/clk/CU of L2 -> LDS traffic (5.8 TB/s chip-wide, 75 % L2 hits),
// the per-CU delivery rate of this access pattern.  Removing work from the MFMA stream no longer helps (21 -> 20
// products: -0.5 %), removing bytes does (column groups, below: -9 %).
#include "gpbo_internal.h"

#include <limits>

namespace {

typedef int i4_t __attribute__((ext_vector_type(4)));
typedef int i16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int NS = 6;                      // int8 slices of U = diagonals a + b kept (int32 accumulator sets)
constexpr int NSA = 5;                     // int8 slices of K*: its sixth digit would only meet U's first (pair (5,0), dropped)
constexpr int BM = 128, BN = 128, BK = 32;  // candidates x columns of V per workgroup, k depth of a stage
constexpr int FRAG = 1024;                 // bytes of one 32 x 32 int8 fragment
constexpr int A_STAGE = (BM / 32) * NSA * FRAG;  // 20 KiB: the K* part of a stage
constexpr int B_STAGE = (BN / 32) * NS * FRAG;   // 24 KiB: the U part
constexpr int STAGE = A_STAGE + B_STAGE;   // 44 KiB
constexpr int A_PIECES = A_STAGE / FRAG, PIECES = STAGE / FRAG;  // 20 + 24 DMA pieces per stage: 6 for waves 0-3, 5 for 4-7
constexpr int KS_SLICE = GPBO_KS_SLICE;    // observations per mu_part slice (64)
constexpr double MAGIC = 6755399441055744.0;  // 1.5 2^52: x + MAGIC has rint(x) in its low mantissa bits
constexpr double TWO38 = 274877906944.0;

struct LsArgsI8 {
    double isc[GPBO_MAX_D];  // 1 / (ls_k sqrt 2)
};

__device__ __forceinline__ void glds16b(const char *g, char *l) {
    __builtin_amdgcn_global_load_lds((glb_void_t *)g, (lds_void_t *)l, 16, 0, 0);
}

// The same DMA with the address as wave-uniform base (SGPR pair) + one 32-bit lane offset and the LDS destination as a
// byte address for M0: no address VGPRs (the builtin form above costs a 64-bit VGPR address per piece, which sigma_i8c_kernel
// cannot afford).  The compiler does not track M0 or the memory counter across this statement: a kernel that uses it issues
// ALL its DMA this way and counts vmcnt itself.
__device__ __forceinline__ void glds16b_s(const char *base, unsigned lane_off, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(lane_off), "s"(base) : "memory");
}

// ---- exp(-t), t >= 0: the one definition shared with kernel_build.hip (bit-identical entries and means) --------------
#include "exp_neg.h"

// ---- balanced base-256 digits of four fixed-point values at once --------------------------------------------------
// In: z[q] = x_q 2^46 + MAGIC (q = 0..3).  Out: P[a] (a = 0 most significant) = the a-th digits of the four values,
// one int8 per byte (value q in byte q).  T' = T + 0x8080808080; digit byte = byte of T' (xor 0x80 below the top).
__device__ __forceinline__ void digits4(const double (&z)[4], unsigned (&P)[NS]) {
    unsigned lo[4], hi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned l = (unsigned)__double2loint(z[q]);
        const unsigned h = (unsigned)__double2hiint(z[q]) - 0x43380000u;  // high word of T (two's complement)
        lo[q] = l + 0x80808080u;
        hi[q] = h + 0x80u + (lo[q] < l ? 1u : 0u);
    }
    // v_perm_b32: result byte k = byte sel_k of {S0:S1}, selector 0-3 = bytes of S1 (second argument), 4-7 = bytes of S0
    const unsigned t01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400u);  // {l0.b0, l1.b0, l0.b1, l1.b1}
    const unsigned t23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400u);
    const unsigned u01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602u);  // {l0.b2, l1.b2, l0.b3, l1.b3}
    const unsigned u23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602u);
    const unsigned h01 = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400u);  // {h0.b0, h1.b0, h0.b1, h1.b1}
    const unsigned h23 = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400u);
    P[5] = __builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u;  // bytes 0 of the four values
    P[4] = __builtin_amdgcn_perm(t23, t01, 0x07060302u) ^ 0x80808080u;
    P[3] = __builtin_amdgcn_perm(u23, u01, 0x05040100u) ^ 0x80808080u;
    P[2] = __builtin_amdgcn_perm(u23, u01, 0x07060302u) ^ 0x80808080u;
    P[1] = __builtin_amdgcn_perm(h23, h01, 0x05040100u) ^ 0x80808080u;
    P[0] = __builtin_amdgcn_perm(h23, h01, 0x07060302u);                // top digit: plain two's complement byte
}

// Five digits (K*): z[q] = k_q 2^38 + MAGIC, T < 2^39, T' = T + 0x80808080; P[0] = top digit (plain), P[1..4] below it.
__device__ __forceinline__ void digits4_k(const double (&z)[4], unsigned (&P)[NSA]) {
    unsigned lo[4], hi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned l = (unsigned)__double2loint(z[q]);
        const unsigned h = (unsigned)__double2hiint(z[q]) - 0x43380000u;
        lo[q] = l + 0x80808080u;
        hi[q] = h + (lo[q] < l ? 1u : 0u);
    }
    const unsigned t01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400u);
    const unsigned t23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400u);
    const unsigned u01 = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602u);
    const unsigned u23 = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602u);
    const unsigned h01 = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400u);
    const unsigned h23 = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400u);
    P[4] = __builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u;
    P[3] = __builtin_amdgcn_perm(t23, t01, 0x07060302u) ^ 0x80808080u;
    P[2] = __builtin_amdgcn_perm(u23, u01, 0x05040100u) ^ 0x80808080u;
    P[1] = __builtin_amdgcn_perm(u23, u01, 0x07060302u) ^ 0x80808080u;
    P[0] = __builtin_amdgcn_perm(h23, h01, 0x05040100u);
}

// ---- U -> column scales and int8 fragments (once per factorisation) ---------------------------------------------------
// scale[j] = 2^e_j >= max_i |U_ij|  (exact power of two), inv[j] = 2^(46 - e_j).
// grid Np/64, block 1024: 64 columns x 16 row groups (a column's rows are strided over the groups, one 512-byte row
// segment per wave load), maximum over the groups through LDS.  (One thread per column walking all its rows took 1 ms at
// N = 4096 - as long as the whole K* slicing of a chunk.)
__global__ __launch_bounds__(1024) void u_colscale_kernel(const double *__restrict__ U, int Np, double *__restrict__ scale,
                                                          double *__restrict__ inv) {
    __shared__ double part[16][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    double m = 0.0;
    for (int i = rg; i <= j; i += 16) m = fmax(m, fabs(U[(int64_t)i * Np + j]));  // upper triangular: rows <= j
    part[rg][c] = m;
    __syncthreads();
    if (rg == 0) {
        for (int q = 1; q < 16; ++q) m = fmax(m, part[q][c]);
        int e = 0;
        if (m > 0.0) (void)frexp(m, &e);  // m = f 2^e, f in [0.5, 1)
        scale[j] = ldexp(1.0, e);
        inv[j] = ldexp(1.0, 46 - e);
    }
}

// thread = (k block of 16, column): 16 entries -> six 16-byte lane operands.  grid (Np/256, Np/16), block 256.
__global__ __launch_bounds__(256) void u_slices_kernel(const double *__restrict__ U, int Np, const double *__restrict__ inv,
                                                       char *__restrict__ U8) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= Np) return;  // Np is a multiple of 128, not of 256
    const int k0 = blockIdx.y * 16;
    const double sc = inv[col];
    unsigned out[NS][4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        double z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + q4 * 4 + q;
            const double u = (k <= col) ? U[(int64_t)k * Np + col] : 0.0;  // zeros below the diagonal, whatever is stored
            z[q] = fma(u, sc, MAGIC);
        }
        unsigned P[NS];
        digits4(z, P);
#pragma unroll
        for (int a = 0; a < NS; ++a) out[a][q4] = P[a];
    }
    const int kb = k0 >> 5, kg = (k0 >> 4) & 1, ct = col >> 5, CT = Np >> 5;
    char *base = U8 + (((int64_t)kb * CT + ct) * NS) * FRAG + kg * 512 + (col & 31) * 16;
#pragma unroll
    for (int a = 0; a < NS; ++a)
        *reinterpret_cast<i4_t *>(base + a * FRAG) = i4_t{(int)out[a][0], (int)out[a][1], (int)out[a][2], (int)out[a][3]};
}

// ---- K(X*,X) chunk -> int8 fragments + fp64 mean partials ---------------------------------------------------------------
// grid (ldk_used/256, Np/64), block 256: thread = one candidate, blockIdx.y = 64 observations (four 16-entry lane
// operands).  Entries and mean partials are the fp64 path's (same distance / exp arithmetic, same fma order over n).
// NA = digits stored per entry: 5 for the full pass, 3 (the leading ones: a balanced representation truncates to nearest)
// for the coarse screen below.
template <int D, bool NT /* non-temporal stores: slabs far beyond the Infinity Cache (see kernel_build.hip) */, int NA>
__global__ __launch_bounds__(256) void kstar_slices_kernel(const double *__restrict__ Xs, int64_t Mc,
                                                           const double *__restrict__ Xsc, int N, LsArgsI8 ls,
                                                           const double *__restrict__ alpha, char *__restrict__ A8,
                                                           int64_t RT, double *__restrict__ mu_part, int64_t ldk) {
    __shared__ double tab[GPBO_EXP_E];
    if (threadIdx.x < GPBO_EXP_E) tab[threadIdx.x] = kExp2Tab256[threadIdx.x * (256 / GPBO_EXP_E)];
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double x[D];
    bool nan_c = false;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        x[k] = ((c < Mc) ? Xs[c * D + k] : 0.0) * ls.isc[k];
        nan_c = nan_c || (x[k] != x[k]);
    }
    __syncthreads();
    double mu = 0.0;
    const int nb = blockIdx.y * KS_SLICE;
#pragma unroll 1
    for (int g16 = 0; g16 < KS_SLICE / 16; ++g16) {
        const int n0 = nb + g16 * 16;
        unsigned out[NA][4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            double z[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + q4 * 4 + q;
                double kv = 0.0;
                if (n < N) {  // wave-uniform
                    const double *xo = Xsc + (int64_t)n * D;
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const double dd = x[k] - xo[k];
                        s = fma(dd, dd, s);
                    }
                    kv = exp_neg(s, tab);
                    mu = fma(kv, alpha[n], mu);
                }
                z[q] = fma(kv, TWO38, MAGIC);
            }
            unsigned P[NSA];
            digits4_k(z, P);
#pragma unroll
            for (int a = 0; a < NA; ++a) out[a][q4] = P[a];
        }
        const int kb = n0 >> 5, kg = (n0 >> 4) & 1;
        char *base = A8 + (((int64_t)kb * RT + (c >> 5)) * NA) * FRAG + kg * 512 + (int)(c & 31) * 16;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const i4_t v = {(int)out[a][0], (int)out[a][1], (int)out[a][2], (int)out[a][3]};
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<i4_t *>(base + a * FRAG));
            else *reinterpret_cast<i4_t *>(base + a * FRAG) = v;
        }
    }
    mu_part[(int64_t)blockIdx.y * ldk + c] = nan_c ? __builtin_nan("") : mu;
}

#ifdef GPBO_I8_STAMPS  // diagnostics build only (tools/build_variant.sh): where do the waves wait?  cycles summed over waves
__device__ unsigned long long g_i8_stamps[4];  // [0] total wave cycles, [1] in the pre-barrier s_waitcnt, [2] in s_barrier, [3] waves
__device__ unsigned long long g_i8_timeline[2][96][8];  // waves 0 and 4 of workgroup 0: stamps of stages 100..195
#define I8_TL(slot)                                                                                         \
    do {                                                                                                    \
        if (tl_on && tl_stage >= 100 && tl_stage < 196 && lane == 0)                                        \
            g_i8_timeline[wid >> 2][tl_stage - 100][slot] = __builtin_amdgcn_s_memtime();                  \
    } while (0)
#else
#define I8_TL(slot) do {} while (0)
#endif

// ---- the variance kernel -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void sigma_i8_kernel(
    const char *__restrict__ A8, int64_t RT, const char *__restrict__ U8, int Np, const double *__restrict__ colscale,
    const double *__restrict__ mu_part, int nsl, int64_t ldk, int64_t Mc, double prior_var, int acq_kind, double p0,
    double p1, int64_t idx_base, double *__restrict__ mu_out, double *__restrict__ sigma_out,
    double *__restrict__ acq_out, double *__restrict__ var_out, double *__restrict__ part_val,
    int64_t *__restrict__ part_idx, unsigned long long *__restrict__ nan_count, int G, int nblk,
    double *__restrict__ ss_part /* G > 1: [G x ldk] partial |v|^2 per column group, epilogue in split_finish_kernel */) {
    __shared__ __attribute__((aligned(16))) char smem[3 * STAGE];
    // Column groups (G > 1).  One workgroup streams its 128 rows of K* once per column block of V: with a workgroup per
    // candidate tile and compute unit, 256 such slabs (3 MB each at N = 4096) are live at a time - far beyond the 256-MiB
    // Infinity Cache, so every re-read comes from HBM (55 GB per 2^17 candidates).  Splitting the column blocks of a tile
    // over G workgroups keeps 256 / G slabs live (G = 8: 96 MB beside the 100 MB of U slices).  The G workgroups of a tile
    // get linear ids 8 apart - the same XCD under the round-robin dispatch (speed only) - and column blocks in
    // boustrophedon rounds (g, 2G-1-g, 2G+g, ...: block jb costs jb + 1 stages, so every group gets the same work).
    int tile = blockIdx.x, grp = 0;
    if (G > 1) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int q = slot / G;
        grp = slot - q * G;
        tile = q * 8 + xcd;
        if (tile >= nblk) return;
    }
    auto jb_of = [&](int r) { return r * G + ((r & 1) ? (G - 1 - grp) : grp); };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid & 1, wq = wid >> 1;  // 2 row groups of 64 candidates x 4 column groups of 32 columns
    const int CT = Np >> 5;
    const int nJ = Np / BN;
    const int lane16 = lane * 16;

    // staging iterator over the flattened stage sequence (jb, kb): kb = 0 .. 4 (jb + 1) - 1 for jb = 0 .. nJ - 1
#ifdef GPBO_I8_DIAG_SAME_A  // timing-only diagnostic (wrong results): every workgroup streams the same K* rows (L2 hits)
    const char *a0p = A8;
#else
    const char *a0p = A8 + ((int64_t)tile * (BM / 32) * NSA) * FRAG;  // k block 0 of this workgroup's row tiles
#endif
    const int64_t a_step = RT * NSA * FRAG, b_step = (int64_t)CT * NS * FRAG;
    int pr = 0, pj = jb_of(0), pk = 0, pbuf = 0;
    const char *pa = a0p, *pb = U8 + ((int64_t)pj * (BN / 32) * NS) * FRAG;
    // one DMA piece of a stage: piece p = wid + 8 q of the 44 (20 of K*, then 24 of U; the LDS image is the same order)
    auto piece = [&](const char *ga, const char *gb, int buf, int q) {
        const int p = wid + 8 * q;
        char *dst = smem + buf * STAGE + p * FRAG;
        if (p < A_PIECES) glds16b(ga + p * FRAG + lane16, dst);
        else if (p < PIECES) glds16b(gb + (p - A_PIECES) * FRAG + lane16, dst);
    };
    auto stage_next = [&]() {
#pragma unroll
        for (int q = 0; q < 6; ++q) piece(pa, pb, pbuf, q);
        pbuf = (pbuf == 2) ? 0 : pbuf + 1;
        if (++pk == (pj + 1) * (BN / BK)) {
            pj = jb_of(++pr);
            pk = 0;
            pa = a0p;
            pb = U8 + ((int64_t)pj * (BN / 32) * NS) * FRAG;
        } else {
            pa += a_step;
            pb += b_step;
        }
    };

    i16_t acc[NS][2];
#pragma unroll
    for (int g = 0; g < NS; ++g)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][t][r] = 0;
    double ssrow = 0.0;  // |v|^2 of ONE row of this wave's tile (which row: see the butterfly below)

    // ---- software pipeline ---------------------------------------------------------------------------------------
    // Slice pairs of a stage in row order: M(i) = { K*_i x U_j : j <= 5 - i }, i = 0..4 (12, 10, 8, 6, 4 MFMAs: 20 pairs).
    // LDS ring, three stages, ONE barrier per stage, placed before M(3):
    //   before it  every wave has all its LDS operands of stage t in registers (K* slices 3 and 4 are fetched early) and
    //              has waited for its own DMA pieces of stage t+1 (issued a whole stage earlier; the pieces of stage
    //              t+2 may still be in flight: counted vmcnt, 6 pieces for waves 0-3, 5 for waves 4-7);
    //   after it   stage t+1 is complete for everyone, so its first operands are fetched under the cover of M(3), M(4)
    //              of stage t - the matrix pipe does not drain at a stage boundary - and the buffer of stage t is free:
    //              it takes the DMA of stage t+3, which the two waves of a SIMD issue at different times (waves 0-3
    //              inside M(0), waves 4-7 inside M(1) / M(2) of the next stage), one piece after every second MFMA.
    const bool early = wid < 4;  // which of the two waves of a SIMD issues its DMA pieces first (and has 6, not 5)
    // static priority for the second-dispatched half, the loser of the SIMD's age-based arbitration (same-box A/B:
    // 71.1 -> 70.7 ms per 2^19 candidates; priority for the first half instead: 71.5)
    if (!early) __builtin_amdgcn_s_setprio(1);
    auto wait_own = [&](int stages_left_in_flight) {  // own pieces of all but the youngest `stages_left_in_flight` stages
        if (stages_left_in_flight >= 1) {
            if (early) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
    };
    int inflight = 0;  // stages issued and not yet waited for
    stage_next();
    ++inflight;
    if (pj < nJ) { stage_next(); ++inflight; }
    wait_own(inflight - 1);
    --inflight;
    __builtin_amdgcn_s_barrier();
    bool dma_due = pj < nJ;   // a stage is waiting to be issued into the free buffer (stage 2 into buffer 2 at first)

#ifdef GPBO_I8_STAMPS
    unsigned long long st_wait = 0, st_bar = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
    const bool tl_on = blockIdx.x == 0 && (wid == 0 || wid == 4);
    int tl_stage = 0;
#endif
    int cur = 0;
    i4_t b[NS], a0[2];       // U slices of the current stage, K* slice 0 of the current stage (both row tiles)
    auto lds_a = [&](i4_t (&dst)[2], int buf, int i) {
        const char *As = smem + buf * STAGE + (2 * wr) * NSA * FRAG + lane16;
        dst[0] = *reinterpret_cast<const i4_t *>(As + i * FRAG);
        dst[1] = *reinterpret_cast<const i4_t *>(As + (NSA + i) * FRAG);
    };
    auto lds_b = [&](int buf, int j) {
        b[j] = *reinterpret_cast<const i4_t *>(smem + buf * STAGE + A_STAGE + (wq * NS + j) * FRAG + lane16);
    };
    const char *da = pa, *db = pb;   // sources of the stage being issued piecewise
    int dbuf = pbuf;
    auto dma_piece = [&](int q) { piece(da, db, dbuf, q); };
    auto dma_begin = [&]() {  // latch the sources of the next stage and advance the iterator
        da = pa; db = pb; dbuf = pbuf;
        pbuf = (pbuf == 2) ? 0 : pbuf + 1;
        if (++pk == (pj + 1) * (BN / BK)) {
            pj = jb_of(++pr);
            pk = 0;
            pa = a0p;
            pb = U8 + ((int64_t)pj * (BN / 32) * NS) * FRAG;
        } else {
            pa += a_step;
            pb += b_step;
        }
    };
#pragma unroll
    for (int j = 0; j < NS; ++j) lds_b(0, j);
    lds_a(a0, 0, 0);

#define MM(av, i, j)                                                                                         \
    do {                                                                                                     \
        acc[(i) + (j)][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[0], b[j], acc[(i) + (j)][0], 0, 0, 0); \
        acc[(i) + (j)][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[1], b[j], acc[(i) + (j)][1], 0, 0, 0); \
    } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)
    auto stage_body = [&](bool active) {
        const int nxt = (cur == 2) ? 0 : cur + 1;
        i4_t a1[2], a2[2], a3[2], a4[2];
        const bool issue_now = dma_due;
        if (issue_now) { dma_begin(); ++inflight; }
        SB();
        I8_TL(0);
        lds_a(a1, cur, 1);
        SB();
        // M(0): j descending - U slice 0 of this stage was the last operand fetched
        if (active) { MM(a0, 0, 5); } SB(); if (issue_now && early) dma_piece(0); SB();
        if (active) { MM(a0, 0, 4); } SB(); if (issue_now && early) dma_piece(1); SB();
        if (active) { MM(a0, 0, 3); } SB(); if (issue_now && early) dma_piece(2); SB();
        if (active) { MM(a0, 0, 2); } SB(); if (issue_now && early) dma_piece(3); SB();
        if (active) { MM(a0, 0, 1); } SB(); if (issue_now && early) dma_piece(4); SB();
        if (active) { MM(a0, 0, 0); } SB(); if (issue_now && early) dma_piece(5); SB();
        I8_TL(1);
        lds_a(a2, cur, 2);
        SB();
        if (active) { MM(a1, 1, 0); } SB(); if (issue_now && !early) dma_piece(0); SB();
        if (active) { MM(a1, 1, 1); } SB(); if (issue_now && !early) dma_piece(1); SB();
        if (active) { MM(a1, 1, 2); } SB(); if (issue_now && !early) dma_piece(2); SB();
        if (active) { MM(a1, 1, 3); } SB(); if (issue_now && !early) dma_piece(3); SB();
        if (active) { MM(a1, 1, 4); }
        SB();
        I8_TL(2);
        lds_a(a3, cur, 3);
        lds_a(a4, cur, 4);
        SB();
        if (active) { MM(a2, 2, 0); } SB(); if (issue_now && !early) dma_piece(4); SB();
        if (active) { MM(a2, 2, 1); MM(a2, 2, 2); MM(a2, 2, 3); }
        SB();
        I8_TL(3);
        // all LDS operands of this stage are in registers; own pieces of the next stage have landed
#ifdef GPBO_I8_STAMPS
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
        wait_own(inflight - 1);
        if (inflight > 0) --inflight;
#ifdef GPBO_I8_STAMPS
        const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef GPBO_I8_STAMPS
        const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
        st_wait += ts1 - ts0;
        st_bar += ts2 - ts1;
        if (tl_on && tl_stage >= 100 && tl_stage < 196 && lane == 0) {
            g_i8_timeline[wid >> 2][tl_stage - 100][4] = ts1;
            g_i8_timeline[wid >> 2][tl_stage - 100][5] = ts2;
        }
#endif
        dma_due = pj < nJ;   // the buffer of this stage is free from here on
        SB();
        // (issuing the pieces of waves 4-7 right here, after the barrier, instead of inside M(1) / M(2) changes nothing:
        //  69.8 against 69.7 ms per 2^19 candidates; letting waves 0-3 issue all 44 pieces, 11 each, LOSES: 71.1 -> 74.3 ms)
        // next stage's operands under the cover of M(3), M(4); registers of dead U slices are reused as they die
        lds_b(nxt, 5); lds_b(nxt, 4); lds_b(nxt, 3);
        lds_a(a0, nxt, 0);
        SB();
        if (active) { MM(a3, 3, 0); MM(a3, 3, 1); MM(a3, 3, 2); }
        SB();
        I8_TL(6);
        lds_b(nxt, 2);
        SB();
        if (active) { MM(a4, 4, 0); MM(a4, 4, 1); }
        SB();
        I8_TL(7);
        lds_b(nxt, 1);
        lds_b(nxt, 0);
        SB();
#ifdef GPBO_I8_STAMPS
        ++tl_stage;
#endif
        cur = nxt;
    };
#undef MM
#undef SB

    for (int rr = 0, jb = jb_of(0); jb < nJ; jb = jb_of(++rr)) {
        const int ct = jb * (BN / 32) + wq;        // this wave's 32-column tile of V
        const int nkb = (jb + 1) * (BN / BK);
        for (int kb = 0; kb < nkb; ++kb) stage_body(kb <= ct);  // U is upper triangular: k tiles below the diagonal are zero
        // column block finished: v = 2^e_j sum_g 2^(-12-8g) G_g, squared and summed over this wave's 32 columns
        const double cs = colscale[ct * 32 + (lane & 31)];
        double x[32];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double v = (double)acc[5][t][r] * 0x1p-52;
                v = fma((double)acc[4][t][r], 0x1p-44, v);
                v = fma((double)acc[3][t][r], 0x1p-36, v);
                v = fma((double)acc[2][t][r], 0x1p-28, v);
                v = fma((double)acc[1][t][r], 0x1p-20, v);
                v = fma((double)acc[0][t][r], 0x1p-12, v);
                v *= cs;
                x[t * 16 + r] = v * v;
#pragma unroll
                for (int g = 0; g < NS; ++g) acc[g][t][r] = 0;
            }
        // butterfly over the 32 lanes of a half wave: 32 values per lane -> 1; lane bits b0..b4 end up holding the
        // column sum of value index Q = 16 b0 + 8 b1 + 4 b2 + 2 b3 + b4
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int half = 16 >> s;
            const bool up = (lane >> s) & 1;
#pragma unroll
            for (int q = 0; q < half; ++q) {
                const double keep = up ? x[q + half] : x[q];
                const double send = up ? x[q] : x[q + half];
                x[q] = keep + __shfl_xor(send, 1 << s);
            }
        }
        ssrow += x[0];
    }

#ifdef GPBO_I8_STAMPS
    if (lane == 0) {
        atomicAdd(&g_i8_stamps[0], __builtin_amdgcn_s_memtime() - st_begin);
        atomicAdd(&g_i8_stamps[1], st_wait);
        atomicAdd(&g_i8_stamps[2], st_bar);
        atomicAdd(&g_i8_stamps[3], 1ull);
    }
#endif
    // ---- row sums to LDS: red[wq][row of the block]
    __syncthreads();
    double *red = reinterpret_cast<double *>(smem);  // [4][BM]
    {
        const int Q = ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
        const int t = Q >> 4, r = Q & 15;
        // 32 x 32 accumulator map: register r of lane l holds row 8 (r >> 2) + 4 (l >> 5) + (r & 3), column l & 31
        const int row = wr * 64 + t * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        red[wq * BM + row] = ssrow;
    }
    __syncthreads();
    if (ss_part) {  // (kernel-argument uniform) column-group launch: partial sums only
        if (tid < BM)
            ss_part[(int64_t)grp * ldk + (int64_t)tile * BM + tid] =
                ((red[tid] + red[BM + tid]) + red[2 * BM + tid]) + red[3 * BM + tid];
        return;
    }
    double *s_val = red + 4 * BM;
    int64_t *s_idx = reinterpret_cast<int64_t *>(red + 4 * BM + 4);
    if (tid < BM) {
        const int64_t c = (int64_t)tile * BM + tid;
        const bool valid = c < Mc;
        const double ssq = ((red[tid] + red[BM + tid]) + red[2 * BM + tid]) + red[3 * BM + tid];
        double mu = 0.0;
        for (int s = 0; s < nsl; ++s) mu += mu_part[(int64_t)s * ldk + c];
        const double var = prior_var - ssq;
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
            if (gpbo_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_val[tid >> 6] = bv; s_idx[tid >> 6] = bi; }
    }
    __syncthreads();
    if (tid == 0) {
        double bv = s_val[0];
        int64_t bi = s_idx[0];
        if (gpbo_better(s_val[1], s_idx[1], bv, bi)) { bv = s_val[1]; bi = s_idx[1]; }
        part_val[tile] = bv;
        part_idx[tile] = bi;
    }
}

// ---- coarse screen: three digits per operand, three diagonals ------------------------------------------------------------
// The screen in front of rescore.hip does not need a 1e-10 variance: the arg-max is decided in fp64 among the candidates
// whose interval [acq(var - tau), acq(var + tau)] reaches the best lower bound, and on the benchmark problem a tau of
// 1e-3 still leaves a handful of candidates per 2^16 (tools/ozaki_error.py + DESIGN 4c).  Keeping the three leading
// digits of both operands (k to 2^-23, U to 2^(e_j - 23): the leading digits of a balanced representation ARE the
// round-to-nearest truncation) and the diagonals a + b <= 2 is SIX slice products instead of twenty:
//     |dsigma^2| <= 1.9e-4 at N = 4096 ("sk=3 su=3 keep=3"), tau is checked on every call like the other screens'.
// Three int32 accumulator sets leave room for a 256 x 128 block tile (wave tile 64 x 64 = 2 x 2 MFMA tiles x 3 = 192
// registers): a stage (32 k) is 24 KiB of K* + 12 KiB of U (slices 0-2 of the six that gpbo_prepare_i8 stores) for
// 6 x 8 x 4 = 192 MFMAs - 18 KiB per 128 x 128 x 32 of the product instead of 44, which is what the full pass is bound by.
// Ring of four stages (144 KiB): two whole stages are in flight behind the one being consumed.
constexpr int CNA = 3;                         // digits of K* the coarse pass stores and reads
constexpr int CND = 3;                         // diagonals kept = digits of U read
constexpr int CBM = 256, CBN = 128;
constexpr int CA_PIECES = (CBM / 32) * CNA;    // 24
constexpr int CB_PIECES = (CBN / 32) * CND;    // 12
constexpr int CPIECES = CA_PIECES + CB_PIECES; // 36: waves 0-3 issue 5 per stage, waves 4-7 issue 4
constexpr int CSTAGE = CPIECES * FRAG;         // 36 KiB
#ifndef GPBO_I8C_RING
#define GPBO_I8C_RING 4
#endif
constexpr int CRING = GPBO_I8C_RING;

__global__ __launch_bounds__(512) void sigma_i8c_kernel(const char *__restrict__ A8, int64_t RT, const char *__restrict__ U8,
                                                        int Np, const double *__restrict__ colscale, int64_t ldk, int G,
                                                        int nblk, double *__restrict__ ss_part /* [G x ldk] */) {
    __shared__ __attribute__((aligned(16))) char smem[CRING * CSTAGE + 2 * FRAG];
    char *scl = smem + CRING * CSTAGE;   // two 1-KiB buffers: the 128 column scales of a column block (by block parity)
    int tile = blockIdx.x, grp = 0;      // column groups exactly as in sigma_i8_kernel
    if (G > 1) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int q = slot / G;
        grp = slot - q * G;
        tile = q * 8 + xcd;
        if (tile >= nblk) return;
    }
    auto jb_of = [&](int r) { return r * G + ((r & 1) ? (G - 1 - grp) : grp); };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid & 3, wq = wid >> 2;   // 4 row groups of 64 candidates x 2 column groups of 64 columns
    const int CT = Np >> 5;
    const int nJ = Np / CBN;
    const int lane16 = lane * 16;
    const bool five = wid < 4;               // this wave has a fifth DMA piece (32 + wid < 36)

#ifdef GPBO_I8C_DIAG_SAME_A   // timing-only diagnostic: every workgroup streams the same K* rows (L2 hits)
    const char *a0p = A8;
#else
    const char *a0p = A8 + ((int64_t)tile * (CBM / 32) * CNA) * FRAG;
#endif
    const int64_t a_step = RT * CNA * FRAG, b_step = (int64_t)CT * NS * FRAG;
    int pr = 0, pj = jb_of(0), pk = 0, pbuf = 0;
    const char *pa = a0p, *pb = U8 + ((int64_t)pj * (CBN / 32) * NS) * FRAG;
    // DMA pieces of a stage: piece p = wid + 8 q of the 36 (24 of K*, then 12 of U; the LDS image is in the same order), so
    // q = 0..2 are always K* pieces, q = 3 is U piece `wid`, q = 4 is U piece 8 + wid for waves 0-3 (no branches on p)
    // (addresses = wave-uniform base + one 32-bit lane offset: the scalar-base form of the DMA instruction, no address VGPRs)
    const int j3 = wid, j4 = 8 + (wid & 3);
    const int boff3 = ((j3 / CND) * NS + (j3 % CND)) * FRAG, boff4 = ((j4 / CND) * NS + (j4 % CND)) * FRAG;
    const unsigned l16 = (unsigned)lane16;
    const unsigned lds0 = (unsigned)(size_t)(lds_void_t *)smem;
#if defined(GPBO_I8C_DIAG_NO_DMA)   // timing-only diagnostics (wrong results; tools/build_variant.sh), never in the shipped library
    constexpr bool kLoopDma = false;
#else
    constexpr bool kLoopDma = true;
#endif
    auto stage_issue = [&](bool dma) {   // all of this wave's pieces of the next stage, then advance the iterator
        if (dma) {
            // first stage of a column block: its 128 column scales (1 KiB) travel the same way, issued BEFORE the stage's
            // pieces, so whoever has waited for the stage has them too.  (A global load in the epilogue would drain the
            // whole DMA queue: the counter is shared.)  The extra piece only makes wave 4's counted waits stricter.
            if (pk == 0 && wid == 4)
                glds16b_s(reinterpret_cast<const char *>(colscale + (int64_t)pj * CBN), l16,
                          lds0 + CRING * CSTAGE + (pr & 1) * FRAG);
            const unsigned dst = lds0 + pbuf * CSTAGE + wid * FRAG;
            const char *paw = pa + wid * FRAG;
            glds16b_s(paw, l16, dst);
            glds16b_s(paw + 8 * FRAG, l16, dst + 8 * FRAG);
            glds16b_s(paw + 16 * FRAG, l16, dst + 16 * FRAG);
            glds16b_s(pb + boff3, l16, dst + 24 * FRAG);
            if (five) glds16b_s(pb + boff4, l16, dst + 32 * FRAG);
        }
        pbuf = (pbuf == CRING - 1) ? 0 : pbuf + 1;
        if (++pk == (pj + 1) * (CBN / BK)) {
            pj = jb_of(++pr);
            pk = 0;
            pa = a0p;
            pb = U8 + ((int64_t)pj * (CBN / 32) * NS) * FRAG;
        } else {
            pa += a_step;
            pb += b_step;
        }
    };
    // own pieces of every stage but the youngest `younger` ones have landed (counted: loads return in order)
    auto wait_own = [&](int younger) {
        if (younger >= 2) {
            if (five) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        } else if (younger == 1) {
            if (five) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
    };

    i16_t acc[CND][2][2];   // [diagonal][row tile][column tile]
#pragma unroll
    for (int g = 0; g < CND; ++g)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][t][u][r] = 0;
    double ssrow = 0.0;

    // ---- software pipeline -----------------------------------------------------------------------------------------
    // The operands of stage t are in registers when its products start; the six products (four MFMAs each) are ordered
    // so that operand registers die early,  (0,2) (0,1) | (1,1) (0,0) (1,0) (2,0),  and each dead fragment pair is
    // refilled with stage t+1's at once - a whole product or more ahead of its first use.  One barrier per stage,
    // after the second product: before it every wave has waited for its own pieces of stage t+1 (issued three stages
    // ago) and has every operand of stage t in registers; after it stage t+1 is complete for everyone and the buffer
    // of stage t is free for the DMA of stage t+4: three whole stages are in flight behind the one being consumed.
    int ahead = 0;   // stages issued beyond the one in registers
    stage_issue(true);
    for (int q = 0; q < CRING - 1; ++q)
        if (pj < nJ) { stage_issue(true); ++ahead; }
    wait_own(ahead);
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    i4_t af[2][CNA], bf[2][CND];
    auto lds_a = [&](int buf, int i) {
        const char *As = smem + buf * CSTAGE + (2 * wr) * CNA * FRAG + lane16;
        af[0][i] = *reinterpret_cast<const i4_t *>(As + i * FRAG);
        af[1][i] = *reinterpret_cast<const i4_t *>(As + (CNA + i) * FRAG);
    };
    auto lds_b = [&](int buf, int j) {
        const char *Bs = smem + buf * CSTAGE + (CA_PIECES + (2 * wq) * CND) * FRAG + lane16;
        bf[0][j] = *reinterpret_cast<const i4_t *>(Bs + j * FRAG);
        bf[1][j] = *reinterpret_cast<const i4_t *>(Bs + (CND + j) * FRAG);
    };
#pragma unroll
    for (int i = 0; i < CNA; ++i) { lds_b(0, i); lds_a(0, i); }

#ifdef GPBO_I8C_DIAG_NO_MFMA   // timing-only diagnostic: one MFMA per product instead of four
#define CMM(i, j, t, u) \
    do { if ((t) + (u) == 0) acc[(i) + (j)][t][u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[t][i] + af[1][i], bf[u][j] + bf[1][j], acc[(i) + (j)][t][u], 0, 0, 0); } while (0)
#else
#define CMM(i, j, t, u) \
    acc[(i) + (j)][t][u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[t][i], bf[u][j], acc[(i) + (j)][t][u], 0, 0, 0)
#endif
    // No skipping of the k tiles below a column tile's diagonal (at most 3 of the 4 (jb + 1) stages of a column block, 2 %
    // of the products): their U digits are stored as zeros, and this kernel is not bound by the MFMAs.
#define CMM4(i, j) do { CMM(i, j, 0, 0); CMM(i, j, 0, 1); CMM(i, j, 1, 0); CMM(i, j, 1, 1); } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)
    int blk = 0;   // column blocks finished by this workgroup (parity = scale buffer)
    for (int rr = 0, jb = jb_of(0); jb < nJ; jb = jb_of(++rr)) {
        const int nkb = (jb + 1) * (CBN / BK);
        for (int kb = 0; kb < nkb; ++kb) {
            const bool has_next = ahead > 0;
            const int nxt = (cur == CRING - 1) ? 0 : cur + 1;
            SB();
            CMM4(0, 2);
            CMM4(0, 1);
            SB();
            wait_own(has_next ? ahead - 1 : 0);
#ifndef GPBO_I8C_DIAG_NO_BARRIER   // timing-only diagnostic (wrong results: stages are read before other waves' pieces have
            // landed): the upper bound of what ANY barrier-free stage hand-off could gain
            __builtin_amdgcn_s_barrier();
#endif
            SB();
            CMM4(1, 1);   // straight after the barrier: nothing but MFMAs between the release and the pipe's next work
            SB();
            // stage t+4 goes into the buffer of stage t, all of this wave's pieces at once.  Same-box A/B, ms per 2^19
            // candidates: here 23.6; before the product above (the pipe idles while every wave issues DMA) 24.5; one piece
            // after each of the remaining products 24.9; the two waves of a SIMD at different times (w: here, w + 4: two
            // products later) 24.4
            if (pj < nJ) stage_issue(kLoopDma);
            else if (ahead > 0) --ahead;
            SB();
            // (after the last stage these fetch a buffer nobody fills any more: harmless, and unconditional loads let the
            //  compiler count lgkmcnt instead of draining it at the loop head)
            lds_b(nxt, 2);
            lds_b(nxt, 1);
            SB();
            CMM4(0, 0);
            SB();
            lds_a(nxt, 0);
            SB();
            CMM4(1, 0);
            SB();
            lds_a(nxt, 1);
            SB();
            CMM4(2, 0);
            SB();
            lds_a(nxt, 2);
            lds_b(nxt, 0);
            SB();
            cur = nxt;
        }
        // column block finished: v = 2^e_j (2^-12 G_0 + 2^-20 G_1 + 2^-28 G_2), squared and summed over the wave's 64
        // columns.  Per row tile: 16 values per lane -> 1 by a butterfly over lane bits 0..3 (level s keeps the half of
        // the values whose register-index bit 3 - s equals lane bit s), then lane bit 4 chooses the row tile.
        const double *sc = reinterpret_cast<const double *>(scl + (blk & 1) * FRAG) + wq * 64 + (lane & 31);
        const double cs0 = sc[0], cs1 = sc[32];
        ++blk;
        double y[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            double x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double v0 = (double)acc[2][t][0][r] * 0x1p-28;
                v0 = fma((double)acc[1][t][0][r], 0x1p-20, v0);
                v0 = fma((double)acc[0][t][0][r], 0x1p-12, v0);
                v0 *= cs0;
                double v1 = (double)acc[2][t][1][r] * 0x1p-28;
                v1 = fma((double)acc[1][t][1][r], 0x1p-20, v1);
                v1 = fma((double)acc[0][t][1][r], 0x1p-12, v1);
                v1 *= cs1;
                x[r] = fma(v1, v1, v0 * v0);
#pragma unroll
                for (int g = 0; g < CND; ++g) { acc[g][t][0][r] = 0; acc[g][t][1][r] = 0; }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int half = 8 >> s;
                const bool up = (lane >> s) & 1;
#pragma unroll
                for (int q = 0; q < half; ++q) {
                    const double keep = up ? x[q + half] : x[q];
                    const double send = up ? x[q] : x[q + half];
                    x[q] = keep + __shfl_xor(send, 1 << s);
                }
            }
            y[t] = x[0];
        }
        {
            const bool up = (lane >> 4) & 1;
            const double keep = up ? y[1] : y[0];
            const double send = up ? y[0] : y[1];
            ssrow += keep + __shfl_xor(send, 16);
        }
    }
#undef CMM4
#undef CMM
#undef SB

    __syncthreads();
    double *red = reinterpret_cast<double *>(smem);  // [2][CBM]
    {
        // the row this lane's sum belongs to: register index r = b0 b1 b2 b3 (lane bits, b0 most significant), row tile b4;
        // 32 x 32 accumulator map: register r of lane l holds row 8 (r >> 2) + 4 (l >> 5) + (r & 3), column l & 31
        const int r = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
        const int t = (lane >> 4) & 1;
        const int row = wr * 64 + t * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        red[wq * CBM + row] = ssrow;
    }
    __syncthreads();
    if (tid < CBM) ss_part[(int64_t)grp * ldk + (int64_t)tile * CBM + tid] = red[tid] + red[CBM + tid];
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct LayoutI8 {
    int64_t a8_off, mup_off, xsc_off, pval_off, pidx_off, nan_off, ssp_off, total, nparts_cap;
};

#ifndef GPBO_I8_G
#define GPBO_I8_G 8
#endif
constexpr int I8_GROUPS_MAX = 16;

// column groups per candidate tile: none for short problems (few column blocks: the fused epilogue is worth more)
int i8_groups(int64_t Np) {
    const int64_t nJ = Np / BN;
    if (nJ < 16) return 1;
    return (GPBO_I8_G <= nJ / 2) ? GPBO_I8_G : (int)(nJ / 2);
}

LayoutI8 layout_i8(int64_t Np, int64_t chunk, int64_t M) {
    LayoutI8 L;
    const int64_t nchunks = (M + chunk - 1) / chunk;
    L.nparts_cap = nchunks * ((chunk + 255) / 256);   // partials of the 128-row fused epilogue or the 256-row split one
    if (L.nparts_cap < nchunks * (chunk / BM)) L.nparts_cap = nchunks * (chunk / BM);
    int64_t off = 0;
    L.a8_off = off; off += align_up(Np * chunk * NSA, 256);
    L.mup_off = off; off += align_up((int64_t)sizeof(double) * (Np / KS_SLICE) * chunk, 256);
    L.xsc_off = off; off += align_up((int64_t)sizeof(double) * Np * GPBO_MAX_D, 256);
    L.pval_off = off; off += align_up((int64_t)sizeof(double) * L.nparts_cap, 256);
    L.pidx_off = off; off += align_up((int64_t)sizeof(int64_t) * L.nparts_cap, 256);
    L.nan_off = off; off += 256;
    L.ssp_off = off; off += align_up((int64_t)sizeof(double) * I8_GROUPS_MAX * chunk, 256);
    L.total = off;
    return L;
}

}  // namespace

#ifdef GPBO_I8_STAMPS
extern "C" int gpbo_i8_timeline_read(unsigned long long *out_host /* [2][96][8] */) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_i8_timeline), sizeof(unsigned long long) * 2 * 96 * 8) == hipSuccess ? 0 : -2;
}
extern "C" int gpbo_i8_stamps_read(unsigned long long *out4_host, int reset) {
    if (hipMemcpyFromSymbol(out4_host, HIP_SYMBOL(g_i8_stamps), 4 * sizeof(unsigned long long)) != hipSuccess) return -2;
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_i8_stamps), z, sizeof(z)) != hipSuccess) return -2;
    }
    return 0;
}
#endif

extern "C" int64_t gpbo_prepare_i8_bytes(int64_t Np) {
    if (Np < GPBO_NPAD || Np % GPBO_NPAD || Np > GPBO_I8_MAX_N) return GPBO_ERR_ARG;
    return align_up(Np * Np * NS, 256) + 2 * align_up((int64_t)sizeof(double) * Np, 256);
}

extern "C" int gpbo_prepare_i8(const double *U, int64_t Np, void *u8, int64_t u8_bytes, void *stream) {
    if (!U || !u8 || ((uintptr_t)u8 & 255)) return GPBO_ERR_ARG;
    const int64_t need = gpbo_prepare_i8_bytes(Np);
    if (need < 0) return GPBO_ERR_ARG;
    if (u8_bytes < need) return GPBO_ERR_WORKSPACE;
    hipStream_t st = gpbo_stream(stream);
    char *U8 = reinterpret_cast<char *>(u8);
    double *scale = reinterpret_cast<double *>(U8 + align_up(Np * Np * NS, 256));
    double *inv = scale + align_up((int64_t)sizeof(double) * Np, 256) / 8;
    hipLaunchKernelGGL(u_colscale_kernel, dim3((unsigned)(Np / 64)), dim3(1024), 0, st, U, (int)Np, scale, inv);
    hipLaunchKernelGGL(u_slices_kernel, dim3((unsigned)(Np / 256 + (Np % 256 ? 1 : 0)), (unsigned)(Np / 16)), dim3(256), 0, st,
                       U, (int)Np, inv, U8);
    GPBO_CHECK_LAUNCH();
    return GPBO_OK;
}

extern "C" int64_t gpbo_posterior_workspace_bytes_i8(int64_t Np, int64_t chunk, int64_t M) {
    if (Np < GPBO_NPAD || Np % GPBO_NPAD || Np > GPBO_I8_MAX_N || chunk < GPBO_CHUNK_GRANULE || chunk % GPBO_CHUNK_GRANULE ||
        chunk > GPBO_CHUNK_MAX || M < 1)
        return GPBO_ERR_ARG;
    return layout_i8(Np, chunk, M).total;
}

namespace {
int posterior_i8_impl(bool coarse, const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                      const double *ls_host, const void *u8, const double *alpha, double prior_var, int32_t acq_kind,
                      double p0, double p1, int64_t idx_offset, int64_t chunk, double *mu_out, double *sigma_out,
                      double *acq_out, double *var_out, gpbo_result *result, void *work, int64_t work_bytes,
                      gpbo_profile *prof, void *stream) {
    if (!Xs || !X || !u8 || !alpha || !result || !work || !ls_host) return GPBO_ERR_ARG;
    if (M < 1 || N < 1 || Np != gpbo_padded_n(N) || Np > GPBO_I8_MAX_N || d < 1 || d > GPBO_MAX_D) return GPBO_ERR_ARG;
    if (chunk < GPBO_CHUNK_GRANULE || chunk % GPBO_CHUNK_GRANULE || chunk > GPBO_CHUNK_MAX) return GPBO_ERR_ARG;
    if (acq_kind != GPBO_ACQ_LCB && acq_kind != GPBO_ACQ_EI) return GPBO_ERR_ARG;
    if (((uintptr_t)work & 255) || ((uintptr_t)u8 & 255)) return GPBO_ERR_ARG;
    const LayoutI8 L = layout_i8(Np, chunk, M);
    if (work_bytes < L.total) return GPBO_ERR_WORKSPACE;
    LsArgsI8 ls;
    for (int k = 0; k < GPBO_MAX_D; ++k) ls.isc[k] = 0.0;
    for (int k = 0; k < d; ++k) {
        if (!(ls_host[k] > 0.0)) return GPBO_ERR_ARG;
        ls.isc[k] = 1.0 / (ls_host[k] * 1.4142135623730950488);
    }
    hipStream_t st = gpbo_stream(stream);
    char *w = reinterpret_cast<char *>(work);
    char *A8 = w + L.a8_off;
    double *mu_part = reinterpret_cast<double *>(w + L.mup_off);
    double *Xsc = reinterpret_cast<double *>(w + L.xsc_off);
    double *part_val = reinterpret_cast<double *>(w + L.pval_off);
    int64_t *part_idx = reinterpret_cast<int64_t *>(w + L.pidx_off);
    unsigned long long *nan_count = reinterpret_cast<unsigned long long *>(w + L.nan_off);
    const char *U8 = reinterpret_cast<const char *>(u8);
    const double *colscale = reinterpret_cast<const double *>(U8 + align_up(Np * Np * NS, 256));
    int rc = gpbo_scale_points_launch(X, N, Np, d, ls_host, Xsc, nan_count, stream);
    if (rc != GPBO_OK) return rc;
    const int64_t RT = chunk / 32;
    int64_t nparts = 0;
    bool prev_recorded = false;
    for (int64_t s = 0; s < M; s += chunk) {
        const int64_t Mc = (M - s < chunk) ? (M - s) : chunk;
        const bool rec = prof && prof->count < prof->capacity;
        if (rec) {
            const bool chained = s > 0 && prof->count > 0 && prev_recorded;
            prof->kmode[prof->count] = chained ? 2 : 1;
            if (!chained && hipEventRecord(reinterpret_cast<hipEvent_t>(prof->kbegin[prof->count]), st) != hipSuccess)
                return GPBO_ERR_LAUNCH;
        }
        // fragments of whole 128-candidate blocks are read by the variance kernel: build them for every block touched
        const int64_t used = (Mc + 255) / 256 * 256;
        dim3 kgrid((unsigned)(used / 256), (unsigned)(Np / KS_SLICE));
        const bool nt = Np * chunk * (coarse ? CNA : NSA) > ((int64_t)1 << 30);
#define CALL1(DD, NTT, NAA)                                                                                               \
    hipLaunchKernelGGL((kstar_slices_kernel<DD, NTT, NAA>), kgrid, dim3(256), 0, st, Xs + s * d, Mc, Xsc, (int)N, ls, alpha,  \
                       A8, RT, mu_part, chunk)
#define CALL(DD)                                                                                                          \
    do {                                                                                                                  \
        if (coarse) { if (nt) CALL1(DD, true, CNA); else CALL1(DD, false, CNA); }                                         \
        else { if (nt) CALL1(DD, true, NSA); else CALL1(DD, false, NSA); }                                                \
    } while (0)
        switch (d) {
            case 1: CALL(1); break;   case 2: CALL(2); break;   case 3: CALL(3); break;   case 4: CALL(4); break;
            case 5: CALL(5); break;   case 6: CALL(6); break;   case 7: CALL(7); break;   case 8: CALL(8); break;
            case 9: CALL(9); break;   case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break;
            case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; case 16: CALL(16); break;
            default: return GPBO_ERR_ARG;
        }
#undef CALL
#undef CALL1
        const int64_t nblk = (Mc + BM - 1) / BM;
        if (rec && hipEventRecord(reinterpret_cast<hipEvent_t>(prof->begin[prof->count]), st) != hipSuccess)
            return GPBO_ERR_LAUNCH;
        const int G = i8_groups(Np);
        int64_t nparts_here = nblk;
        if (coarse) {   // 256-row tiles, partial sums always finished by split_finish_kernel
            double *ss_part = reinterpret_cast<double *>(w + L.ssp_off);
            const int64_t nblk_c = (Mc + CBM - 1) / CBM;
            const int64_t grid = (G > 1) ? (nblk_c + 7) / 8 * 8 * G : nblk_c;
            hipLaunchKernelGGL(sigma_i8c_kernel, dim3((unsigned)grid), dim3(512), 0, st, A8, RT, U8, (int)Np, colscale, chunk, G,
                               (int)nblk_c, ss_part);
            nparts_here = nblk_c;
            int rc2 = gpbo_launch_split_finish(ss_part, G, chunk, mu_part, (int)(Np / KS_SLICE), Mc, prior_var, acq_kind, p0, p1,
                                               idx_offset + s, mu_out ? mu_out + s : nullptr, sigma_out ? sigma_out + s : nullptr,
                                               acq_out ? acq_out + s : nullptr, var_out ? var_out + s : nullptr,
                                               part_val + nparts, part_idx + nparts, nan_count, st);
            if (rc2 != GPBO_OK) return rc2;
        } else if (G > 1) {
            double *ss_part = reinterpret_cast<double *>(w + L.ssp_off);
            const int64_t grid = (nblk + 7) / 8 * 8 * G;
            hipLaunchKernelGGL(sigma_i8_kernel, dim3((unsigned)grid), dim3(512), 0, st, A8, RT, U8, (int)Np, colscale, mu_part,
                               (int)(Np / KS_SLICE), chunk, Mc, prior_var, (int)acq_kind, p0, p1, idx_offset + s,
                               (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr, part_val + nparts,
                               part_idx + nparts, nan_count, G, (int)nblk, ss_part);
            nparts_here = (Mc + 255) / 256;
            int rc2 = gpbo_launch_split_finish(ss_part, G, chunk, mu_part, (int)(Np / KS_SLICE), Mc, prior_var, acq_kind, p0, p1,
                                               idx_offset + s, mu_out ? mu_out + s : nullptr, sigma_out ? sigma_out + s : nullptr,
                                               acq_out ? acq_out + s : nullptr, var_out ? var_out + s : nullptr,
                                               part_val + nparts, part_idx + nparts, nan_count, st);
            if (rc2 != GPBO_OK) return rc2;
        } else {
            hipLaunchKernelGGL(sigma_i8_kernel, dim3((unsigned)nblk), dim3(512), 0, st, A8, RT, U8, (int)Np, colscale, mu_part,
                               (int)(Np / KS_SLICE), chunk, Mc, prior_var, (int)acq_kind, p0, p1, idx_offset + s,
                               mu_out ? mu_out + s : nullptr, sigma_out ? sigma_out + s : nullptr,
                               acq_out ? acq_out + s : nullptr, var_out ? var_out + s : nullptr, part_val + nparts,
                               part_idx + nparts, nan_count, 1, (int)nblk, (double *)nullptr);
        }
        if (rec) {
            if (hipEventRecord(reinterpret_cast<hipEvent_t>(prof->end[prof->count]), st) != hipSuccess)
                return GPBO_ERR_LAUNCH;
            prof->cands[prof->count] = Mc;
            ++prof->count;
        }
        prev_recorded = rec;
        GPBO_CHECK_LAUNCH();
        nparts += nparts_here;
    }
    return gpbo_launch_argmax_finish(part_val, part_idx, nparts, nan_count, result, st);
}
}  // namespace

extern "C" int gpbo_posterior_acq_i8(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                                     const double *ls_host, const void *u8, const double *alpha, double prior_var,
                                     int32_t acq_kind, double p0, double p1, int64_t idx_offset, int64_t chunk,
                                     double *mu_out, double *sigma_out, double *acq_out, double *var_out,
                                     gpbo_result *result, void *work, int64_t work_bytes, gpbo_profile *prof,
                                     void *stream) {
    return posterior_i8_impl(false, Xs, M, X, N, Np, d, ls_host, u8, alpha, prior_var, acq_kind, p0, p1, idx_offset, chunk,
                             mu_out, sigma_out, acq_out, var_out, result, work, work_bytes, prof, stream);
}

extern "C" int gpbo_posterior_acq_i8c(const double *Xs, int64_t M, const double *X, int64_t N, int64_t Np, int32_t d,
                                      const double *ls_host, const void *u8, const double *alpha, double prior_var,
                                      int32_t acq_kind, double p0, double p1, int64_t idx_offset, int64_t chunk,
                                      double *mu_out, double *sigma_out, double *acq_out, double *var_out,
                                      gpbo_result *result, void *work, int64_t work_bytes, gpbo_profile *prof,
                                      void *stream) {
    return posterior_i8_impl(true, Xs, M, X, N, Np, d, ls_host, u8, alpha, prior_var, acq_kind, p0, p1, idx_offset, chunk,
                             mu_out, sigma_out, acq_out, var_out, result, work, work_bytes, prof, stream);
}
