"""How many first-level survivors the exact bound would have if its mean came from an fp32 exponential (VERDICT round 4,
item 4): the bound's own acq_ub of every candidate of the headline problem (and config 4's), and the count of candidates whose
acq_ub + delta still reaches the best exact value, for slacks delta of the mean.  python tools/bound_fp32_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_optimisation_amd import DeviceGP, _lib
from bayesian_optimisation_amd.synthetic import ard_length_scales, rff_objective, sobol_points

for (N, d, M) in ((4096, 8, 1 << 21), (8192, 16, 1 << 19)):
    ls = ard_length_scales(d)
    X = sobol_points(0, N, d); y = rff_objective(X, ls); Xs = sobol_points(N, M, d)
    gp = DeviceGP().factorise(X, y, ls, order="fps")
    S0 = float(gp.alpha[:N].abs().sum())
    for name, kw, kind, p0 in (("lcb4", dict(acquisition="lcb", explore=4.0), _lib.ACQ_LCB, 4.0),
                               ("ei", dict(acquisition="ei", f_best=float(y.min())), _lib.ACQ_EI, float(y.min()))):
        best = gp.score(Xs, **kw).best_val
        J = gp.bound_prefix()
        Xd = gp._dev(Xs)
        chunk, wbytes = gp._ensure_post_workspace(M)
        o = [torch.empty(M, dtype=torch.float64, device=gp.device) for _ in range(3)]
        st = gp.lib.gpbo_posterior_prefix_f64(gp._ptr(Xd), M, gp._ptr(gp.X), gp.N, gp.Np, gp.d, gp.ls_h.ctypes.data_as(C.c_void_p),
                                              gp._ptr(gp.U), gp._ptr(gp.alpha), 1.000101, kind, p0, 0.0, 0, chunk, J,
                                              gp._ptr(o[0]), gp._ptr(o[1]), gp._ptr(o[2]), gp._ptr(gp._result),
                                              gp._ptr(gp._work_post), wbytes, None, gp._stream())
        assert st == 0
        mu_lb, sig_ub = o[0], o[1]
        line = [f"N={N} d={d} M={M} {name}: S0 = sum|alpha| = {S0:.3g}, 1.5e-7 S0 = {1.5e-7 * S0:.3g}; prefix {J}; survivors at slack"]
        for delta in (0.0, 1e-6, 1e-4, 1e-3, 3e-3, 1e-2, 1.5e-7 * S0, 3e-2, 1e-1):
            if kind == _lib.ACQ_LCB:
                ub = p0 * sig_ub - (mu_lb - delta)
                n = int((ub >= best).sum())
            else:   # EI rises by at most delta when the mean falls by delta
                n = int((o[2] + delta >= best).sum())
            line.append(f"{delta:.2g}: {n}")
        print(" | ".join(line), flush=True)
    del gp
    torch.cuda.empty_cache()
