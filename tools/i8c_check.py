# coarse int8 screen (three digits per operand, gpbo_posterior_acq_i8c): accuracy against the fp64 kernels, decision against
# the fp64 kernels, then timing at the headline shape beside the full int8 screen and the fp64 pass.
# usage: python tools/i8c_check.py [time]
import ctypes as C
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP, _lib
from bayesian_optimisation_amd.gp_device import PRIOR_VAR
from bayesian_optimisation_amd.synthetic import make_problem

for N, M, d, chunk in [(100, 1000, 3, 512), (256, 2048, 8, 1024), (700, 5000, 8, 2048), (2048, 4096, 8, 4096),
                       (2500, 70000, 8, 1 << 15), (4096, 1 << 16, 8, 1 << 16)]:
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP(chunk=chunk).factorise(X, y, ls)
    for acq_kw in (dict(acquisition="lcb"), dict(acquisition="ei", f_best=float(y.min()))):
        r = gp.score_i8c(Xs, dense=True, idx_offset=3, **acq_kw)
        scr = dict(gp.last_screen)
        r64 = gp.score(Xs, dense=True, idx_offset=3, **acq_kw)
        v8 = r.sigma.cpu().numpy() ** 2
        v64 = r64.sigma.cpu().numpy() ** 2
        print(N, M, d, acq_kw["acquisition"], "dmu", np.abs(r.mu.cpu().numpy() - r64.mu.cpu().numpy()).max(),
              "dvar %.3g" % np.abs(v8 - v64).max(), "dsigma %.3g" % np.abs(r.sigma.cpu().numpy() - r64.sigma.cpu().numpy()).max(),
              "idx", r.best_idx == r64.best_idx, "val", r.best_val == r64.best_val, scr, flush=True)

if len(sys.argv) > 1:
    N, M, d = 4096, 1 << 19, 8
    X, y, Xs, ls = make_problem(N, M, d)
    gp = DeviceGP().factorise(X, y, ls)
    Xd = gp._dev(Xs)

    def raw(fnname):
        def run(P):
            if not getattr(gp, "_u8_valid", False):
                gp.prepare_i8()
            m = int(P.shape[0])
            need = int(gp.lib.gpbo_posterior_workspace_bytes_i8(gp.Np, gp.chunk, m))
            if getattr(gp, "_w8", None) is None:
                gp._w8 = torch.empty(need // 8 + 1, dtype=torch.float64, device=gp.device)
                gp._v8 = torch.empty(m, dtype=torch.float64, device=gp.device)
            st = getattr(gp.lib, fnname)(gp._ptr(P), m, gp._ptr(gp.X), gp.N, gp.Np, gp.d, gp.ls_h.ctypes.data_as(C.c_void_p),
                                         gp._ptr(gp.U8), gp._ptr(gp.alpha), PRIOR_VAR, 0, 4.0, 0.0, 0, gp.chunk, None, None,
                                         None, gp._ptr(gp._v8), gp._ptr(gp._result), gp._ptr(gp._w8), need, None, gp._stream())
            _lib.check(st, fnname)
            v, i, n = gp.read_result(gp._result)

            class R:
                best_idx = i
                best_val = v
            return R
        return run

    for name, fn in [("i8c", gp.score_i8c), ("i8c pass alone", raw("gpbo_posterior_acq_i8c")), ("i8", gp.score_i8),
                     ("i8 pass alone", raw("gpbo_posterior_acq_i8")), ("f64", gp.score)]:
        fn(Xd)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3):
            r = fn(Xd)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        print(name, "%.1f ms per 2^19" % (dt * 1e3), "%.4g cand/s" % (M / dt), r.best_idx, r.best_val,
              gp.last_screen if name in ("i8c", "i8") else "", flush=True)
