// Launch plan of the fused Cholesky + inverse factor (cholinv.hip).  Host-only C++ (no HIP types): the same header is
// compiled into libgpbo and into the CPU simulator tests/c/cholinv_sim.cpp, which executes the plan with plain loops
// and checks it against LAPACK - so the schedule (what reads what, in which launch) is verified without a GPU.
//
// What is factorised: cov_meas = K(X,X) + jitter (/root/reference/point_selector.py:79), which the reference inverts
// with np.linalg.inv (:89).  Here S = [A | W] is one Np x 2Np row-major matrix, A = cov_meas, W = 0.  ROW operations
// L^-1 [A | I] = [L^T | L^-1] are applied block row by block row (64 rows), so that on exit
//     A (upper block triangle) = R = L^T         W (lower triangle) = L^-1 = U^T
// and the variance kernel's U = L^-T is one transposition away.  Every product of the algorithm has the shape
//     out[r, c] -= sum_k S[k, r] * S[k, c]        (k = finished rows; r = target rows; c = live columns of those rows)
// with both operands read k-major (row k contiguous) - the layout the gfx950 fp64 MFMA fragments want from LDS.
// The live columns of block row b are contiguous in S: [64 b, Np) of A, then [Np, Np + wlim) of W.
//
// Schedule.  Rows are taken in pairs of block rows (128 rows).  PAIR(p) factorises the pair's 128 x 128 diagonal block
// (redundantly in every workgroup of the launch) and multiplies the pair's rows by the inverse of the block's factor.
// The pair's rank-128 contribution then goes to the NEXT pair's rows at once (NEAR) and to all later rows (FAR) as
// filler workgroups inside the next pair's launch: one stream, dependencies by launch order only, no events, no
// in-kernel flags.
#pragma once
#include <stdint.h>

#include <vector>

enum { CI_NONE = 0, CI_PAIR = 1, CI_UPD_SMALL = 2, CI_UPD_BIG = 3 };

struct CiJob {
    int32_t kind;   // CI_*
    int32_t nblk;   // workgroups of this job in the launch
    int32_t j;      // PAIR: pair index p (rows [128 p, 128 p + 128))
    int32_t k0, K;  // UPD: source rows [k0, k0 + K)
    int32_t r0, r1; // UPD: target rows [r0, r1), multiples of 64
    int32_t wlim;   // UPD: live W columns [Np, Np + wlim)
    int32_t t0;     // UPD: first tile of the job's enumeration taken by this launch
};

struct CiLaunch {
    CiJob job[3];
};

#if defined(__HIPCC__)
#define CI_HD __host__ __device__
#else
#define CI_HD
#endif

// ---- tile enumerations (shared by the planner, the kernels and the simulator) -------------------------------------
// SMALL: 64 x 64 tiles that cover the live region exactly.  Block row b has (Np + wlim) / 64 - b tiles.
CI_HD inline int ci_small_ntiles(int Np, int r0, int r1, int wlim) {
    const int tot = (Np + wlim) / 64;
    int n = 0;
    for (int b = r0 / 64; b < r1 / 64; ++b) n += tot - b;
    return n;
}
CI_HD inline void ci_small_decode(int Np, int r0, int wlim, int t, int *row0, int *col0) {
    const int tot = (Np + wlim) / 64;
    int b = r0 / 64;
    while (t >= tot - b) { t -= tot - b; ++b; }
    *row0 = 64 * b;
    *col0 = 64 * (b + t);
}
// BIG: 128 x 128 tiles on 128-aligned columns of S; row tile i starts at r0 + 128 i (r0 a multiple of 64); stores are
// masked to the live region (64-granular), rows to r1.
CI_HD inline int ci_big_ncol(int Np, int rr, int wlim) {
    return (Np + (wlim + 127) / 128 * 128 - rr / 128 * 128) / 128;
}
CI_HD inline int ci_big_ntiles(int Np, int r0, int r1, int wlim) {
    int n = 0;
    for (int rr = r0; rr < r1; rr += 128) n += ci_big_ncol(Np, rr, wlim);
    return n;
}
CI_HD inline void ci_big_decode(int Np, int r0, int wlim, int t, int *row0, int *col0) {
    int rr = r0;
    for (;;) {
        const int nc = ci_big_ncol(Np, rr, wlim);
        if (t < nc) break;
        t -= nc;
        rr += 128;
    }
    *row0 = rr;
    *col0 = rr / 128 * 128 + 128 * t;
}
// PAIR(p): one workgroup per 64 live columns of the pair's rows: (Np - 128 (p + 1)) / 64 of A, 128 p / 64 of W, and the
// two column blocks of W's own diagonal block = Np / 64 for every p.
CI_HD inline int ci_pair_ntiles(int Np) { return Np / 64; }

struct CiPlanOptions {
    int near_big_from;  // NEAR goes to 128 x 128 tiles when it has more than this many 64 x 64 tiles
};

inline CiPlanOptions ci_default_options(int Np) {
    (void)Np;
    CiPlanOptions o;
    o.near_big_from = 1 << 30;
    return o;
}

inline CiJob ci_upd_job(int kind, int Np, int k0, int K, int r0, int r1, int wlim) {
    CiJob u = {};
    u.kind = kind;
    u.k0 = k0; u.K = K; u.r0 = r0; u.r1 = r1; u.wlim = wlim; u.t0 = 0;
    u.nblk = (kind == CI_UPD_SMALL) ? ci_small_ntiles(Np, r0, r1, wlim) : ci_big_ntiles(Np, r0, r1, wlim);
    return u;
}

// Per pair p: NEAR(p) brings the pair's rows up to date with pair p-1 (the older pairs reached them as FAR fillers),
// then PAIR(p) runs with the FAR tiles of pair p-1 (its contribution to every row beyond pair p) as filler workgroups.
inline std::vector<CiLaunch> ci_plan(int Np, const CiPlanOptions &o) {
    std::vector<CiLaunch> out;
    const int np = Np / 128;
    for (int p = 0; p < np; ++p) {
        if (p > 0) {  // NEAR: rows of pair p  -=  contribution of pair p-1
            CiLaunch l = {};
            const int nsmall = ci_small_ntiles(Np, 128 * p, 128 * p + 128, 128 * p);
            l.job[0] = ci_upd_job(nsmall > o.near_big_from ? CI_UPD_BIG : CI_UPD_SMALL, Np, 128 * (p - 1), 128, 128 * p,
                                  128 * p + 128, 128 * p);
            out.push_back(l);
        }
        CiLaunch l = {};
        l.job[0].kind = CI_PAIR;
        l.job[0].j = p;
        l.job[0].nblk = ci_pair_ntiles(Np);
        // FAR of pair p-1: rows beyond pair p (which NEAR(p) has just served): reads rows of pair p-1 (final), writes
        // rows >= 128 (p + 1) - nothing PAIR(p) touches
        if (p > 0 && 128 * (p + 1) < Np) l.job[2] = ci_upd_job(CI_UPD_BIG, Np, 128 * (p - 1), 128, 128 * (p + 1), Np, 128 * p);
        out.push_back(l);
    }
    return out;
}
