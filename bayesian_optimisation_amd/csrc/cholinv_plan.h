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
// Schedule.  Block rows are taken in groups of G.  Inside a group, row j first receives the contributions of the
// group's earlier rows (NARROW, K = 64 (j - j0)), then PANEL(j) factorises its 64 x 64 diagonal block (redundantly in
// every workgroup of the launch) and multiplies the row by the inverse of the block's factor.  After the group, its
// rank-64G contribution goes to the NEXT group's rows at once (NEAR) and to all later rows (FAR) as filler workgroups
// inside the next group's launches: one stream, dependencies by launch order only, no events, no in-kernel flags.
#pragma once
#include <stdint.h>

#include <vector>

enum { CI_NONE = 0, CI_PANEL = 1, CI_UPD_SMALL = 2, CI_UPD_BIG = 3 };

struct CiJob {
    int32_t kind;   // CI_*
    int32_t nblk;   // workgroups of this job in the launch
    int32_t j;      // PANEL: block row
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
// PANEL(j): 128-wide column tiles of S from the tile that holds column 64 j to the one that holds W's block j
CI_HD inline int ci_panel_ntiles(int Np, int j) {
    const int first = (64 * j) / 128, last = (Np + 64 * j + 63) / 128;
    return last - first + 1;
}

struct CiPlanOptions {
    int G;             // block rows per group (even; the last group may be shorter)
    int near_big_from; // NEAR goes to 128 x 128 tiles when it has more than this many 64 x 64 tiles
    int w_panel, w_narrow;  // share of the FAR filler tiles a panel / narrow launch takes (relative weights)
};

inline CiPlanOptions ci_default_options(int Np) {
    CiPlanOptions o;
    o.G = (Np >= 6144) ? 4 : 2;
    o.near_big_from = 600;
    o.w_panel = 3;
    o.w_narrow = 1;
    return o;
}

inline CiJob ci_upd_job(int kind, int Np, int k0, int K, int r0, int r1, int wlim) {
    CiJob u = {};
    u.kind = kind;
    u.k0 = k0; u.K = K; u.r0 = r0; u.r1 = r1; u.wlim = wlim; u.t0 = 0;
    u.nblk = (kind == CI_UPD_SMALL) ? ci_small_ntiles(Np, r0, r1, wlim) : ci_big_ntiles(Np, r0, r1, wlim);
    return u;
}

inline std::vector<CiLaunch> ci_plan(int Np, const CiPlanOptions &o) {
    std::vector<CiLaunch> out;
    const int nb = Np / 64, G = o.G;
    CiJob far = {};  // the previous group's update of the rows beyond the current group's successor
    for (int j0 = 0; j0 < nb; j0 += G) {
        const int gend = (j0 + G < nb) ? j0 + G : nb;
        std::vector<CiLaunch> L;
        std::vector<int> weight;
        for (int j = j0; j < gend; ++j) {
            if (j > j0) {
                CiLaunch l = {};
                l.job[0] = ci_upd_job(CI_UPD_SMALL, Np, 64 * j0, 64 * (j - j0), 64 * j, 64 * j + 64, 64 * j);
                L.push_back(l);
                weight.push_back(o.w_narrow);
            }
            CiLaunch l = {};
            l.job[0].kind = CI_PANEL;
            l.job[0].j = j;
            l.job[0].nblk = ci_panel_ntiles(Np, j);
            L.push_back(l);
            weight.push_back(o.w_panel);
        }
        // FAR tiles of the previous group: spread over this group's panel / narrow launches (all of them read rows
        // the previous group has finished and write rows beyond this group - nothing this group's launches touch)
        if (far.kind != CI_NONE && far.nblk > 0) {
            int wsum = 0;
            for (int w : weight) wsum += w;
            int done = 0, acc = 0;
            for (size_t i = 0; i < L.size(); ++i) {
                acc += weight[i];
                const int upto = (i + 1 == L.size()) ? far.nblk : (int)((int64_t)far.nblk * acc / wsum);
                if (upto > done) {
                    CiJob f = far;
                    f.t0 = done;
                    f.nblk = upto - done;
                    L[i].job[2] = f;
                    done = upto;
                }
            }
        }
        far = CiJob{};
        if (gend < nb) {
            const int k0 = 64 * j0, K = 64 * (gend - j0), wlim = 64 * gend;
            const int n0 = 64 * gend, n1 = (64 * (gend + G) < Np) ? 64 * (gend + G) : Np;
            CiLaunch l = {};
            const int nsmall = ci_small_ntiles(Np, n0, n1, wlim);
            l.job[0] = ci_upd_job(nsmall > o.near_big_from ? CI_UPD_BIG : CI_UPD_SMALL, Np, k0, K, n0, n1, wlim);
            L.push_back(l);
            if (n1 < Np) far = ci_upd_job(CI_UPD_BIG, Np, k0, K, n1, Np, wlim);
        }
        for (const CiLaunch &l : L) out.push_back(l);
    }
    return out;
}
