"""Randomised sweep of the drop-in classes (ARD grid search included) against the oracle's restatement of
update_surrogate() + lower_confidence_bound(): d in {1, 2}, N = 1 .. 60, random grids of length scales, random
candidate grids, both classes (tensor-resident and host-pointer), fp64; plus the q = 8 qEI and the fp32 scoring
path on random d <= 16 problems.  usage: python tools/fuzz_dropin.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (before the first host-pointer call: a process that uses both routes imports PyTorch first)

from bayesian_optimisation_amd import DeviceGP, PointSelector, PointSelectorHost
from oracle import gp_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
t_end = time.time() + budget
n_cases = n_fail = 0
verbose = bool(os.environ.get("FUZZ_VERBOSE"))


def fail(tag, msg):
    global n_fail
    n_fail += 1
    print(f"FAIL {tag}: {msg}", flush=True)


while time.time() < t_end:
    mode = rng.choice(["dropin", "dropin", "qei", "f32"])
    n_cases += 1
    try:
        if mode == "dropin":
            d = int(rng.integers(1, 3))
            N = int(rng.choice([1, 2, rng.integers(3, 12), rng.integers(12, 60)]))
            g = int(rng.integers(5, 45))
            lo, hi = float(rng.uniform(0, 5)), float(rng.uniform(10, 400))
            if d == 1:
                axis = np.linspace(lo, hi, g)
                Xs, fd = axis.reshape(-1, 1), [g]
                X = rng.choice(axis, N).reshape(-1, 1)
                length_scales = np.linspace(hi / 60, hi * 1.2, int(rng.integers(3, 40)))
            else:
                a0, a1 = np.linspace(lo, hi, g), np.linspace(lo / 2, hi * 2, g)
                Xs = np.stack(np.meshgrid(a0, a1, indexing="ij"), -1).reshape(-1, 2)
                fd = [g, g]
                X = np.stack([rng.choice(a0, N), rng.choice(a1, N)], 1)
                n1, n2 = int(rng.integers(3, 30)), int(rng.integers(3, 30))
                length_scales = [np.linspace(hi / 50, hi, n1), np.linspace(hi / 25, 2 * hi, n2)]
                if n1 == n2:
                    length_scales = np.array(length_scales)
            y = float(rng.choice([1.0, 1e3, 1e6])) * np.abs(rng.standard_normal(N)) + float(rng.choice([0.0, 1e4]))
            cls = PointSelector if rng.random() < 0.5 else PointSelectorHost
            tag = f"{cls.__name__} d={d} N={N} g={g}"
            if verbose:
                print("case", n_cases, tag, flush=True)
            want = O.select_next(X, y, Xs, fd, length_scales=length_scales, route="chol")
            ps = cls()
            ps.name, ps.iteration = "T", 0
            ps.measured_pts, ps.measured_vals = X, y
            ps.feature_domain, ps.predicted_pts, ps.length_scales = fd, Xs, length_scales
            ps.update_surrogate()
            idx = ps.lower_confidence_bound()
            ys = max(1.0, float(np.abs(y).max()))
            if not np.array_equal(np.ravel(ps.kernel_params), np.ravel(want["kernel_params"])):
                # the float32 grid decides the winner: only a tie within float32 rounding may differ
                nl, wl = ps.nlogml.ravel(), want["nlogml"].ravel()
                fin = np.isfinite(wl)
                if not (np.array_equal(np.isfinite(nl), fin) and np.allclose(nl[fin], wl[fin], rtol=3e-6)):
                    fail(tag, f"likelihood grid differs: kernel_params {np.ravel(ps.kernel_params)} vs {np.ravel(want['kernel_params'])}")
                continue  # different cell within rounding: the posterior is not comparable
            e_mu = float(np.max(np.abs(ps.mean_func - want["mean_func"]))) / ys
            e_sig = float(np.max(np.abs(ps.cov_func - want["cov_func"])))
            a = want["acq_func_eval"].ravel()
            top2 = np.sort(a)[-2:] if a.size > 1 else np.array([-np.inf, a[0]])
            if e_mu > 1e-9 or e_sig > 1e-8:
                fail(tag, f"dmu={e_mu:.3g} dsigma={e_sig:.3g}")
            elif top2[1] - top2[0] > 1e-7 * ys and not np.array_equal(idx, want["index"]):
                fail(tag, f"index {idx} vs {want['index']}")
        elif mode == "qei":
            d = int(rng.integers(1, 17))
            N = int(rng.integers(2, 300))
            B = int(rng.integers(1, 300))
            ls = np.exp(rng.uniform(np.log(0.1), np.log(2.0), d))
            X, Xs = rng.uniform(0, 1, (N, d)), rng.uniform(0, 1, (8 * B, d))
            y = rng.standard_normal(N)
            S = int(rng.choice([16, 128, 512]))
            Z = O.qei_base_samples(S, 8, int(rng.integers(0, 100)))
            tag = f"qei d={d} N={N} batches={B} S={S}"
            if verbose:
                print("case", n_cases, tag, flush=True)
            want = O.qei_mc(X, y, Xs, ls, Z, f_best=float(y.min()), xi=0.0)
            gp = DeviceGP(chunk=int(rng.choice([512, 1024, 4096]))).factorise(X, y, ls)
            r = gp.score_qei(Xs, Z, float(y.min()), dense=True)
            got = r.acq.cpu().numpy()
            if np.max(np.abs(got - want)) > 1e-7:
                fail(tag, f"max |dqEI| = {np.max(np.abs(got - want)):.3g}")
            elif r.best_idx != int(np.flatnonzero(got == got.max())[0]):
                fail(tag, "reported batch is not the first maximum of the dense values")
        else:
            d = int(rng.integers(1, 17))
            N = int(rng.integers(1, 500))
            M = int(rng.integers(1, 5000))
            ls = np.exp(rng.uniform(np.log(0.15), np.log(2.0), d))
            X, Xs = rng.uniform(0, 1, (N, d)), rng.uniform(0, 1, (M, d))
            y = rng.standard_normal(N)
            tag = f"f32 d={d} N={N} M={M}"
            if verbose:
                print("case", n_cases, tag, flush=True)
            mu_o, sig_o = O.posterior_chol(X, y, Xs, ls)
            acq_o = O.lcb(mu_o, sig_o, 4)
            gp = DeviceGP().factorise(X, y, ls)
            quirk = 1e-4 if Xs.shape == X.shape else 0.0  # point_selector.py:173
            ys = max(1.0, float(np.abs(y).max()))
            top2 = np.sort(acq_o)[-2:] if M > 1 else np.array([-np.inf, acq_o[0]])
            for route, var_tol in (("score_f32", 5e-3), ("score_i8", 1e-8), ("score_i8c", 1e-3)):
                # screened modes (round 2): the mean is the fp64 kernels' (1e-9 |y| against the oracle as everywhere), the
                # variance carries the screen's error, the selected point is decided in fp64
                r = getattr(gp, route)(Xs, dense=True, diag_add=quirk)
                mu, sig = r.mu.cpu().numpy(), r.sigma.cpu().numpy()
                if np.max(np.abs(mu - mu_o)) > 2e-9 * ys or np.max(np.abs(sig ** 2 - sig_o ** 2)) > var_tol:
                    fail(tag + " " + route, f"dmu={np.max(np.abs(mu - mu_o)):.3g} dvar={np.max(np.abs(sig ** 2 - sig_o ** 2)):.3g}")
                elif r.nan_count or (top2[1] - top2[0] > 1e-7 * ys and r.best_idx != int(np.flatnonzero(acq_o == acq_o.max())[0])):
                    fail(tag + " " + route, "selected point differs from the oracle's first arg-max")
    except Exception as exc:  # noqa: BLE001
        fail(mode, f"{type(exc).__name__}: {exc}")
    if n_cases % 50 == 0:
        print(f"... {n_cases} cases, {n_fail} failures", flush=True)

print(f"fuzz_dropin: {n_cases} cases, {n_fail} failures")
sys.exit(1 if n_fail else 0)
