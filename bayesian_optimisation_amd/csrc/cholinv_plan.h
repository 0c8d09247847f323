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
//           spend as long in their read-modify-write as in their MFMAs).  A far target has no deadline - the window
//           rule catches up with whatever rank is pending - so each launch takes the number of pending far targets that
//           leaves its compute units fullest when its last workgroup ends (every launch ends with a global barrier).
#pragma once
#include <stdint.h>

#include <vector>

enum { CI_NONE = 0, CI_PAIR = 1, CI_UPD_SMALL = 2, CI_UPD_BIG = 3, CI_UPD_BIG256 = 4 };

// out[rows row0.., cols col0..] -= sum_{k in [k0, k0+K)} S[k][row0 + m] * S[k][col0 + n], stores masked to rows < r1 and
// to the live columns (col < Np: col >= 64 floor(row / 64);  col >= Np: col < Np + wlim).  SMALL: 64 x 64 (never masked),
// BIG: 128 x 128, BIG256: 256 x 128.  w: width of a SMALL tile, 64 or 32 (0 = 64).
struct CiTile {
    int32_t kind, k0, K, row0, col0, r1, wlim, w;
};

// workgroups [0, npair) = PAIR(pair) (column tile = workgroup index), then ceil(ntile / group) workgroups that take
// `group` consecutive tiles each of [tile0, tile0 + ntile) of the table (a workgroup that runs several tiles pays the
// dispatch of a 512-thread, 156-KiB-LDS workgroup once: ~5 us per tile at one tile per workgroup)
struct CiLaunch {
    int32_t pair, npair, tile0, ntile, group;
};
inline int ci_launch_blocks(const CiLaunch &l) { return l.npair + (l.ntile + l.group - 1) / l.group; }

struct CiPlan {
    std::vector<CiLaunch> launches;
    std::vector<CiTile> tiles;
};

struct CiPlanOptions {
    int win;       // targets p+1 .. p+win are kept up to date with every finished pair
    int far_k;     // rank of the updates of the targets beyond the window (multiple of 128)
    int far_kind;  // CI_UPD_BIG or CI_UPD_BIG256 (pairs of targets with equal history) for them
    int defer;     // 1: the farther half of a FAR wave waits for the next launch;  2: as many whole targets of the wave
                   // as make the launch's predicted utilisation highest (list scheduling of the cost model on ncu
                   // compute units), the rest later;  3: as many as keep the launch's predicted time within an even
                   // share of all the filler work of the factorisation (a dry run counts it) - the same load on every launch
    int ncu;       // compute units the packer plans for
    int group_from;  // launches with at least this many tiles per compute unit give each workgroup two tiles (0: never)
    int small_w;     // width of the NEAR tiles: 64, or 32 (twice as many workgroups, each half as long: NEAR is on the critical path)
};

inline CiPlanOptions ci_default_options(int Np) {
    CiPlanOptions o;
    o.win = 1;
    // measured on MI355X (end of round 3): N = 8192: 9.10 ms at 384 against 9.26 / 9.28 at 256 / 512; 7168: 6.65 either way;
    // 6144: 4.50 at 256 against 4.68 at 384; 5120: 3.05 against 3.29; 4096: 1.86 against 2.06 (and 2.48 at 128)
    o.far_k = (Np >= 7680) ? 384 : 256;
    o.far_kind = CI_UPD_BIG;
    o.defer = 2;
    o.ncu = 256;
    o.small_w = (Np >= 6144) ? 64 : 32;  // measured: 32 gains 4-5 % at N <= 4096 (NEAR is on the critical path), costs 2 % at 8192
    o.group_from = 0;  // measured on MI355X: two tiles per workgroup cost N = 8192 7 % (coarser tail), gain nothing elsewhere
    return o;
}

inline bool ci_options_ok(const CiPlanOptions &o) {
    return o.win >= 1 && o.win <= 64 && o.far_k >= 128 && o.far_k <= 1024 && o.far_k % 128 == 0 &&
           (o.far_kind == CI_UPD_BIG || o.far_kind == CI_UPD_BIG256) && o.defer >= 0 && o.defer <= 3 && o.ncu >= 1 &&
           o.ncu <= 4096 && o.group_from >= 0 && (o.small_w == 32 || o.small_w == 64);
}

// opt: NULL or int32[7] {win, far_k, far_kind, defer + 1, (max_launches: not a plan option), group_from + 1, small_w};
// 0 keeps the default
inline CiPlanOptions ci_options_from(int Np, const int32_t *opt) {
    CiPlanOptions o = ci_default_options(Np);
    if (opt) {
        if (opt[0] > 0) o.win = opt[0];
        if (opt[1] > 0) o.far_k = opt[1];
        if (opt[2] > 0) o.far_kind = opt[2];
        if (opt[3] > 0) o.defer = opt[3] - 1;
        if (opt[5] > 0) o.group_from = opt[5] - 1;
        if (opt[6] > 0) o.small_w = opt[6];
    }
    return o;
}

// Cost model of the packer (microseconds of one workgroup with the whole chip busy, tools/bench_ci_jobs.py on MI355X).
#ifndef GPBO_CI_COST_PAIR
#define GPBO_CI_COST_PAIR 37.0
#endif
inline double ci_cost_pair() { return GPBO_CI_COST_PAIR; }
inline double ci_cost_tile(int kind, int K) {
    if (kind == CI_UPD_SMALL) return 4.0 + 0.03 * K;
    if (kind == CI_UPD_BIG) return 14.0 + 0.122 * K;
    return 21.0 + 0.244 * K;
}

// Greedy list scheduling, the way the dispatcher hands workgroups to compute units as they free up.
struct CiListSched {
    std::vector<double> heap;  // min-heap of the units' free times
    double work = 0.0, makespan = 0.0;
    explicit CiListSched(int ncu) : heap((size_t)ncu, 0.0) {}
    void add(double cost) {
        // pop min, push min + cost (binary heap by hand: no <algorithm> needed in the device pass)
        double t = heap[0] + cost;
        size_t i = 0, n = heap.size();
        for (;;) {
            size_t l = 2 * i + 1, r = l + 1, m = i;
            double best = t;
            if (l < n && heap[l] < best) { best = heap[l]; m = l; }
            if (r < n && heap[r] < best) { best = heap[r]; m = r; }
            if (m == i) break;
            heap[i] = heap[m];
            i = m;
        }
        heap[i] = t;
        work += cost;
        if (t > makespan) makespan = t;
    }
};

// PAIR(p): one workgroup per 64 live columns of the pair's rows: (Np - 128 (p + 1)) / 64 of A, 128 p / 64 of W, and the
// two column blocks of W's own diagonal block = Np / 64 for every p.
inline int ci_pair_ntiles(int Np) { return Np / 64; }

// tiles of one update: target rows [row0, row0 + rows), sources [k0, k0 + K)
inline void ci_emit_update(std::vector<CiTile> &out, int kind, int Np, int row0, int rows, int k0, int K, int small_w = 64) {
    const int wlim = k0 + K;  // W columns beyond the sources' own extent are zero in the source rows
    if (kind == CI_UPD_SMALL) {
        for (int b = row0 / 64; b < (row0 + rows) / 64; ++b)
            for (int c = 64 * b; c < Np + wlim; c += small_w)
                out.push_back(CiTile{kind, k0, K, 64 * b, c, row0 + rows, wlim, small_w});
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
    double share = 0.0;  // defer == 3: the predicted time a launch may take
    if (o.defer == 3 && np > 2) {
        CiPlanOptions dry = o;
        dry.defer = 0;
        const CiPlan D = ci_plan(Np, dry);
        double fill = 0.0;
        for (const CiTile &t : D.tiles)
            if (t.kind != CI_UPD_SMALL) fill += ci_cost_tile(t.kind, t.K);
        const double per_launch = fill / (np - 2) + ci_pair_ntiles(Np) * ci_cost_pair();
        share = per_launch / o.ncu;
        const double floor_us = ci_cost_pair() + 9.0;  // one round of rank-256 tiles beside the PAIR workgroups
        if (share < floor_us) share = floor_us;
    }
    for (int p = 0; p < np; ++p) {
        if (p > 0) {  // NEAR: whatever pair p still lacks (128 rows when the fillers kept up)
            CiLaunch l = {-1, 0, (int)P.tiles.size(), 0, 1};
            ci_emit_update(P.tiles, CI_UPD_SMALL, Np, 128 * p, 128, applied[p], 128 * p - applied[p], o.small_w);
            applied[p] = 128 * p;
            l.ntile = (int)P.tiles.size() - l.tile0;
            P.launches.push_back(l);
        }
        CiLaunch l = {p, ci_pair_ntiles(Np), (int)P.tiles.size(), 0, 1};
        const int fin = 128 * p;  // source rows below this are final when this launch starts
        // FAR first (long tiles first): targets beyond the window, up to the last multiple of far_k
        const int qb = fin / o.far_k * o.far_k;
        struct FarItem { int t, rows, kind; };
        std::vector<FarItem> far;
        for (int t = p + 1 + o.win; t < np; ++t) {
            if (qb - applied[t] < o.far_k) continue;
            const bool two = o.far_kind == CI_UPD_BIG256 && t + 1 < np && qb - applied[t + 1] >= o.far_k &&
                             applied[t + 1] == applied[t];
            far.push_back(FarItem{t, two ? 256 : 128, two ? CI_UPD_BIG256 : CI_UPD_BIG});
            if (two) ++t;
        }
        if (o.defer >= 2) {
            // oldest debt first (a target that waited enters the window with all of it: a long tile in a launch that
            // should be short), nearer targets first among equals; insertion sort: the list is short and nearly sorted
            for (size_t i = 1; i < far.size(); ++i) {
                const FarItem f = far[i];
                const int pend = qb - applied[f.t];
                size_t j = i;
                while (j > 0 && (qb - applied[far[j - 1].t]) < pend) { far[j] = far[j - 1]; --j; }
                far[j] = f;
            }
        }
        size_t take = far.size();
        if (o.defer == 1 && fin == qb && far.size() > 4) {
            // a fresh wave (the launch right after a far_k boundary): the nearer targets now, the farther ones with the
            // next launch; cost of a target ~ its live width
            int64_t tot = 0, acc = 0;
            for (const FarItem &f : far) tot += (int64_t)(2 * Np - 128 * f.t) * f.rows;
            take = 0;
            while (take < far.size() && 2 * acc < tot) { acc += (int64_t)(2 * Np - 128 * far[take].t) * far[take].rows; ++take; }
        }
        if (o.defer >= 2 && !far.empty()) {
            // Every launch ends with a global barrier, so what counts is how full the compute units are when the last
            // workgroup ends.  Simulate the launch for every prefix of the pending far targets (PAIR workgroups first,
            // then the far tiles, then the window's) and keep the prefix with the best utilisation; nothing is lost by
            // waiting - a far target has no deadline (the window rule catches up with whatever rank is pending).
            std::vector<CiTile> tmp;
            CiListSched cur(o.ncu);
            for (int i = 0; i < l.npair; ++i) cur.add(ci_cost_pair());
            std::vector<double> win_costs;
            for (int t = p + 1; t <= p + o.win && t < np; ++t)
                if (fin - applied[t] >= 128) {
                    tmp.clear();
                    ci_emit_update(tmp, CI_UPD_BIG, Np, 128 * t, 128, applied[t], fin - applied[t]);
                    for (const CiTile &tl : tmp) win_costs.push_back(ci_cost_tile(tl.kind, tl.K));
                }
            double best_u = -1.0, allowed = share * 1.02;
            size_t best_take = 0;
            for (size_t n = 0; n <= far.size(); ++n) {
                if (n > 0) {
                    const FarItem &f = far[n - 1];
                    tmp.clear();
                    ci_emit_update(tmp, f.kind, Np, 128 * f.t, f.rows, applied[f.t], o.far_k);
                    for (const CiTile &tl : tmp) cur.add(ci_cost_tile(tl.kind, tl.K));
                }
                CiListSched end = cur;
                for (double c : win_costs) end.add(c);
                if (o.defer == 3) {
                    // the longest prefix inside the share - or inside what the launch takes anyway without any of them
                    if (n == 0 && end.makespan > allowed) allowed = end.makespan;
                    if (end.makespan <= allowed) best_take = n;
                    else break;
                } else {
                    const double u = end.work / (end.makespan * o.ncu);
                    if (u >= best_u) { best_u = u; best_take = n; }
                }
            }
            take = best_take;
        }
        for (size_t i = 0; i < take; ++i) {
            const FarItem &f = far[i];
            // far_k source rows at a time: a target that waited longer comes back in a later launch (bounded tile length)
            ci_emit_update(P.tiles, f.kind, Np, 128 * f.t, f.rows, applied[f.t], o.far_k);
            applied[f.t] += o.far_k;
            if (f.rows == 256) applied[f.t + 1] += o.far_k;
        }
        // WINDOW: the next targets, fully up to date
        for (int t = p + 1; t <= p + o.win && t < np; ++t) {
            if (fin - applied[t] >= 128) {
                ci_emit_update(P.tiles, CI_UPD_BIG, Np, 128 * t, 128, applied[t], fin - applied[t]);
                applied[t] = fin;
            }
        }
        l.ntile = (int)P.tiles.size() - l.tile0;
        if (o.group_from > 0 && l.ntile >= o.group_from * o.ncu) l.group = 2;
        P.launches.push_back(l);
    }
    return P;
}
