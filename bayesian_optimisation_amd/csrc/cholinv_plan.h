// Launch plan of the fused Cholesky + inverse factor (cholinv.hip).  Host-only C++ (no HIP types): the plan is data - a
// list of launches and a table of tiles - that libgpbo uploads once per matrix size and that tests/cholinv_sim.py
// executes with NumPy (gpbo_cholinv_plan exports it), so the schedule (what reads what, in which launch) is verified
// without a GPU.
//
// What is factorised: cov_meas = K(X,X) + jitter (/root/reference/point_selector.py:79), which the reference inverts
// with np.linalg.inv (:89).  Here S = [A | W] is one Np x 2Np row-major matrix, A = cov_meas, W = 0.  ROW operations
// L^-1 [A | I] = [L^T | L^-1] are applied pair by pair of block rows (128 rows), so that on exit
//     A (upper block triangle) = R = L^T         W (lower triangle) = L^-1 = U^T
// and the variance kernel's U = L^-T is one transposition away.  Every product of the algorithm has the shape
//     out[r, c] -= sum_k S[k, r] * S[k, c]        (k = finished rows; r = target rows; c = live columns of those rows)
// with both operands read k-major (row k contiguous) - the layout the gfx950 fp64 MFMA fragments want from LDS.
// The live columns of block row b are contiguous in S: [64 b, Np) of A, then [Np, Np + wlim) of W.
//
// Schedule.  PAIR(p) factorises the pair's 128 x 128 diagonal block (redundantly in every workgroup of the launch) and
// multiplies the pair's rows by the inverse of the block's factor.  Each pair of target rows t keeps a count applied[t]
// of the source rows it has received; it must be 128 p when PAIR(p = t) runs.  Updates ride as filler workgroups beside
// the PAIR workgroups of a launch (one stream, dependencies by launch order only, no events, no in-kernel flags):
//   NEAR    its own launch before PAIR(p): the last 128 source rows (pair p-1) into pair p, 64 x 64 tiles
//   WINDOW  beside PAIR(p): targets p+1 .. p+win brought up to 128 p (rank 128, or what a deferral left)
//   FAR     beside PAIR(p): targets beyond the window brought up to the last multiple of far_k (rank 256: a rank-128
//           update moves 16 bytes of its target per 256 flop, which is the chip's HBM balance - measured: its tiles
//           spend as long in their read-modify-write as in their MFMAs); half of a FAR wave is deferred to the next
//           launch so that odd and even launches carry similar loads.
#pragma once
#include <stdint.h>

#include <vector>

enum { CI_NONE = 0, CI_PAIR = 1, CI_UPD_SMALL = 2, CI_UPD_BIG = 3, CI_UPD_BIG256 = 4 };

// out[rows row0.., cols col0..] -= sum_{k in [k0, k0+K)} S[k][row0 + m] * S[k][col0 + n], stores masked to rows < r1 and
// to the live columns (col < Np: col >= 64 floor(row / 64);  col >= Np: col < Np + wlim).  SMALL: 64 x 64 (never masked),
// BIG: 128 x 128, BIG256: 256 x 128.
struct CiTile {
    int32_t kind, k0, K, row0, col0, r1, wlim, pad;
};

// workgroups [0, npair) = PAIR(pair) (column tile = workgroup index), then tiles [tile0, tile0 + ntile) of the table
struct CiLaunch {
    int32_t pair, npair, tile0, ntile;
};

struct CiPlan {
    std::vector<CiLaunch> launches;
    std::vector<CiTile> tiles;
};

struct CiPlanOptions {
    int win;       // targets p+1 .. p+win are kept up to date with every finished pair
    int far_k;     // rank of the updates of the targets beyond the window (multiple of 128)
    int far_kind;  // CI_UPD_BIG or CI_UPD_BIG256 (pairs of targets with equal history) for them
    int defer;     // 1: the farther half of a FAR wave waits for the next launch
};

inline CiPlanOptions ci_default_options(int Np) {
    (void)Np;
    CiPlanOptions o;
    o.win = 2;
    o.far_k = 256;
    o.far_kind = CI_UPD_BIG;
    o.defer = 1;
    return o;
}

inline bool ci_options_ok(const CiPlanOptions &o) {
    return o.win >= 1 && o.win <= 64 && o.far_k >= 128 && o.far_k <= 1024 && o.far_k % 128 == 0 &&
           (o.far_kind == CI_UPD_BIG || o.far_kind == CI_UPD_BIG256) && (o.defer == 0 || o.defer == 1);
}

// opt: NULL or int32[4] {win, far_k, far_kind, defer + 1}; 0 keeps the default
inline CiPlanOptions ci_options_from(int Np, const int32_t *opt) {
    CiPlanOptions o = ci_default_options(Np);
    if (opt) {
        if (opt[0] > 0) o.win = opt[0];
        if (opt[1] > 0) o.far_k = opt[1];
        if (opt[2] > 0) o.far_kind = opt[2];
        if (opt[3] > 0) o.defer = opt[3] - 1;
    }
    return o;
}

// PAIR(p): one workgroup per 64 live columns of the pair's rows: (Np - 128 (p + 1)) / 64 of A, 128 p / 64 of W, and the
// two column blocks of W's own diagonal block = Np / 64 for every p.
inline int ci_pair_ntiles(int Np) { return Np / 64; }

// tiles of one update: target rows [row0, row0 + rows), sources [k0, k0 + K)
inline void ci_emit_update(std::vector<CiTile> &out, int kind, int Np, int row0, int rows, int k0, int K) {
    const int wlim = k0 + K;  // W columns beyond the sources' own extent are zero in the source rows
    if (kind == CI_UPD_SMALL) {
        for (int b = row0 / 64; b < (row0 + rows) / 64; ++b)
            for (int c = 64 * b; c < Np + wlim; c += 64) out.push_back(CiTile{kind, k0, K, 64 * b, c, row0 + rows, wlim, 0});
        return;
    }
    const int th = (kind == CI_UPD_BIG256) ? 256 : 128;
    for (int rr = row0; rr < row0 + rows; rr += th)
        for (int c = rr / 128 * 128; c < Np + (wlim + 127) / 128 * 128; c += 128)
            out.push_back(CiTile{kind, k0, K, rr, c, row0 + rows, wlim, 0});
}

inline CiPlan ci_plan(int Np, const CiPlanOptions &o) {
    CiPlan P;
    const int np = Np / 128;
    std::vector<int> applied(np, 0);
    for (int p = 0; p < np; ++p) {
        if (p > 0) {  // NEAR: whatever pair p still lacks (128 rows when the fillers kept up)
            CiLaunch l = {-1, 0, (int)P.tiles.size(), 0};
            ci_emit_update(P.tiles, CI_UPD_SMALL, Np, 128 * p, 128, applied[p], 128 * p - applied[p]);
            applied[p] = 128 * p;
            l.ntile = (int)P.tiles.size() - l.tile0;
            P.launches.push_back(l);
        }
        CiLaunch l = {p, ci_pair_ntiles(Np), (int)P.tiles.size(), 0};
        const int fin = 128 * p;  // source rows below this are final when this launch starts
        // FAR first (long tiles first): targets beyond the window, up to the last multiple of far_k
        const int qb = fin / o.far_k * o.far_k;
        std::vector<int> far;
        for (int t = p + 1 + o.win; t < np; ++t)
            if (qb - applied[t] >= o.far_k) far.push_back(t);
        size_t take = far.size();
        if (o.defer && fin == qb && far.size() > 4) {
            // a fresh wave (the launch right after a far_k boundary): the nearer targets now, the farther ones with the
            // next launch; cost of a target ~ its live width
            int64_t tot = 0, acc = 0;
            for (int t : far) tot += 2 * Np - 128 * t;
            take = 0;
            while (take < far.size() && 2 * acc < tot) acc += 2 * Np - 128 * far[take++];
        }
        for (size_t i = 0; i < take; ++i) {
            const int t = far[i];
            const bool pair_ok = o.far_kind == CI_UPD_BIG256 && i + 1 < take && far[i + 1] == t + 1 &&
                                 applied[t + 1] == applied[t];
            if (pair_ok) {
                ci_emit_update(P.tiles, CI_UPD_BIG256, Np, 128 * t, 256, applied[t], qb - applied[t]);
                applied[t] = applied[t + 1] = qb;
                ++i;
            } else {
                ci_emit_update(P.tiles, CI_UPD_BIG, Np, 128 * t, 128, applied[t], qb - applied[t]);
                applied[t] = qb;
            }
        }
        // WINDOW: the next targets, fully up to date
        for (int t = p + 1; t <= p + o.win && t < np; ++t) {
            if (fin - applied[t] >= 128) {
                ci_emit_update(P.tiles, CI_UPD_BIG, Np, 128 * t, 128, applied[t], fin - applied[t]);
                applied[t] = fin;
            }
        }
        l.ntile = (int)P.tiles.size() - l.tile0;
        P.launches.push_back(l);
    }
    return P;
}
