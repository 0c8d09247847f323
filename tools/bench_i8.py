# timing of the int8-sliced screen at the headline shape (N=4096, d=8), 2^19 candidates, beside the fp64 kernels
# modes: i8 (screen + fp64 decision), f64 (plain fp64 pass), i8raw (the int8 pass alone: gpbo_posterior_acq_i8 through ctypes),
#        i8c / i8craw (the coarse three-digit screen, gpbo_posterior_acq_i8c)
import ctypes as C
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP, _lib
from bayesian_optimisation_amd.gp_device import PRIOR_VAR
from bayesian_optimisation_amd.synthetic import make_problem
N, M, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(__import__("os").environ.get("BENCH_M", 1 << 19)), 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["i8", "f64"]
X, y, Xs, ls = make_problem(N, M, d)
gp = DeviceGP().factorise(X, y, ls)
Xd = gp._dev(Xs)


def i8raw(P, entry="gpbo_posterior_acq_i8"):
    if not getattr(gp, "_u8_valid", False):
        gp.prepare_i8()
        import os
        if os.environ.get("ZERO_U8"):   # power / clock diagnostic: same instruction stream on all-zero U digits
            gp.U8[: gp.Np * gp.Np * 6] = 0
    m = int(P.shape[0])
    need = int(gp.lib.gpbo_posterior_workspace_bytes_i8(gp.Np, gp.chunk, m))
    if getattr(gp, "_w8", None) is None:
        gp._w8 = torch.empty(need // 8 + 1, dtype=torch.float64, device=gp.device)
        gp._v8 = torch.empty(m, dtype=torch.float64, device=gp.device)
    st = getattr(gp.lib, entry)(gp._ptr(P), m, gp._ptr(gp.X), gp.N, gp.Np, gp.d, gp.ls_h.ctypes.data_as(C.c_void_p),
                                      gp._ptr(gp.U8), gp._ptr(gp.alpha), PRIOR_VAR, 0, 4.0, 0.0, 0, gp.chunk, None, None, None,
                                      gp._ptr(gp._v8), gp._ptr(gp._result), gp._ptr(gp._w8), need, None, gp._stream())
    _lib.check(st, "i8raw")
    v, i, n = gp.read_result(gp._result)

    class R:
        best_idx = i
    return R


for name in modes:
    fn = {"i8": gp.score_i8, "f64": gp.score, "i8raw": i8raw, "i8c": gp.score_i8c,
          "i8craw": lambda P: i8raw(P, "gpbo_posterior_acq_i8c"), "bound": gp.score_bound}[name]
    fn(Xd)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        r = fn(Xd)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    print(name, "N=%d %.2f ms per 2^19" % (N, dt * 1e3), "%.4g cand/s" % (M / dt), r.best_idx, flush=True)

if hasattr(gp.lib, "gpbo_i8_stamps_read"):   # diagnostics build (-DGPBO_I8_STAMPS): where the waves of sigma_i8_kernel wait
    buf = (C.c_ulonglong * 4)()
    gp.lib.gpbo_i8_stamps_read(buf, 1)
    i8raw(Xd)
    torch.cuda.synchronize()
    gp.lib.gpbo_i8_stamps_read(buf, 0)
    tot, w, b, n = [int(v) for v in buf]
    print(f"stamps over {n} waves: wave lifetime {tot/n:.0f} cycles, pre-barrier s_waitcnt {100*w/tot:.1f} %, s_barrier {100*b/tot:.1f} %")
    tl = (C.c_ulonglong * (2 * 96 * 8))()
    gp.lib.gpbo_i8_timeline_read(tl)
    T = np.array(list(tl), dtype=np.int64).reshape(2, 96, 8)
    # per stage: [0] body start, [1] after M(0), [2] after M(1), [3] after M(2), [4] after s_waitcnt, [5] after s_barrier,
    #            [6] after M(3), [7] after M(4)
    for w in range(2):
        d = np.diff(T[w], axis=1)
        nxt = T[w, 1:, 0] - T[w, :-1, 7]
        print(f"wave {4*w}: median cycles  M(0) {np.median(d[:,0]):.0f}  M(1) {np.median(d[:,1]):.0f}  M(2) {np.median(d[:,2]):.0f}  "
              f"waitcnt {np.median(d[:,3]):.0f}  barrier {np.median(d[:,4]):.0f}  M(3) {np.median(d[:,5]):.0f}  M(4) {np.median(d[:,6]):.0f}  "
              f"to next stage {np.median(nxt):.0f}  stage {np.median(T[w,1:,0]-T[w,:-1,0]):.0f}")
    print("wave 4 start minus wave 0 start per stage (median):", np.median(T[1,:,0] - T[0,:,0]))
