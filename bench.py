#!/usr/bin/env python3
"""bench.py - candidate acquisitions/sec of the GP acquisition path on MI355X.

One "step" = one BO inner-loop pass over one batch of synthetic input, everything resident in HBM
when the timed region starts:
    factorise K(X,X)+jitter (kxx, Cholesky, U = L^-T, alpha)            [once per step]
    K(X*,X) + mu + sigma + acquisition (LCB, explore=4) + arg-max        [every candidate of the rank]
    one all-gather of (best value, lowest index, NaN count) across ranks [N > 1 only]
Workload at N=1: BASELINE.json configs[1]  (d=8, N=512, M=2^20 Sobol candidates, fp64).
For N>1 the candidate set grows with N (2^20 per GPU, contiguous shards): weak scaling; the reported
value is the whole-job rate M_total / max-over-ranks step time.

Usage: python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = fp32 vector peak
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense fp64 matrix peak = fp64 vector peak (half the 157.3 TF fp32 rate
#                               listed in MI355X_MICROARCH.md; AMD data sheet value)


def cpu_baseline(X, y, Xs_sample, ls):
    """The oracle's Cholesky route (NumPy/LAPACK, BLAS threads = host cores) on a bounded sample of the
    same workload; factorisation excluded (it is amortised over the 2^20 candidates of a real step)."""
    import numpy as np

    from oracle import gp_oracle as O

    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:  # noqa: BLE001
        threads = os.cpu_count() or 1
    try:
        threads = min(threads, len(os.sched_getaffinity(0)))  # BLAS threads that can actually run
    except Exception:  # noqa: BLE001
        pass
    _, L, alpha = O.factorise(X, y, ls)
    t0 = time.perf_counter()
    mu, sig = O.posterior_chol(X, y, Xs_sample, ls, L=L, alpha=alpha)
    acq = O.lcb(mu, sig, 4)
    idx = int(np.flatnonzero(acq == acq.max())[0])
    dt = time.perf_counter() - t0
    return dict(value=len(Xs_sample) / dt, unit="candidate acquisitions/s", cores=int(threads), kind="port",
                sample=f"first {len(Xs_sample)} of the 2^20 candidates, N=512, d=8, posterior+LCB+argmax, "
                       f"{dt:.1f} s wall, factorisation excluded"), idx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n-obs", type=int, default=512)
    ap.add_argument("--m-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=1 << 19,
                    help="candidates timed on the host for cpu_baseline (about 10-20 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="process-group backend; gloo only to rehearse N>1 on one GPU")
    ap.add_argument("--all-on-device", type=int, default=-1,
                    help="rehearsal only: put every rank on this GPU index instead of LOCAL_RANK")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="bracket the kernels with HIP events in every k-th timed step only (an event record costs a few "
                         "microseconds of idle GPU; 1 = every step)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="diagnostic: do not bracket the kernels with HIP events (no roofline in the output)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="rehearsal only: initialise the process group and run the exchange step even at N=1")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                    help="f32: fp64 factorisation, fp32 K*/mean/variance (BASELINE configs[3] shape)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev_index = local_rank if args.all_on_device < 0 else args.all_on_device
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_pg = world > 1 or args.force_process_group
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from bayesian_optimisation_amd import DeviceGP
    from bayesian_optimisation_amd import distributed as D
    from bayesian_optimisation_amd.synthetic import ard_length_scales, rff_objective, sobol_points

    N, d = args.n_obs, args.d
    M_total = args.m_per_gpu * world
    lo, hi = D.shard_bounds(M_total, world, rank)
    ls = ard_length_scales(d)
    X = sobol_points(0, N, d)
    y = rff_objective(X, ls)
    Xs_local = sobol_points(N + lo, hi - lo, d)  # this rank's contiguous shard of the Sobol candidate stream
    kw = dict(chunk=args.chunk) if args.chunk else {}
    gp = DeviceGP(dev, **kw)
    Xd, yd, Xsd = gp._dev(X), gp._dev(y), gp._dev(Xs_local)  # inputs resident in HBM before timing
    if not args.no_kernel_events:
        gp.enable_profile(8192)

    f32 = args.dtype == "f32"

    def score_async():
        if f32:
            return gp.score_async_f32(Xsd, acquisition="lcb", explore=4.0, idx_offset=lo)
        return gp.score_async(Xsd, acquisition="lcb", explore=4.0, idx_offset=lo)

    def step(sample_events=True):
        gp.profile_active = sample_events
        gp.factorise(Xd, yd, ls, check=False)
        if f32:
            gp.prepare_f32()
        score_async()
        # the one exchange step: the 40-byte device record (result + factorisation info) is gathered over the ranks
        # and read back once (at N=1: just the read-back, which synchronises this rank)
        v, i, n, info = D.allreduce_status(gp.status, force_collective=args.force_process_group)
        if info != 0:
            raise RuntimeError("Cholesky failed")
        return v, i, n

    def fence():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        best = step()
    if not args.no_kernel_events:
        gp.reset_profile()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        best = step(i % max(args.event_stride, 1) == 0)
    fence()
    dt = time.perf_counter() - t0
    gp.profile_active = True
    if use_pg:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_step = dt / args.steps * 1e3
    value = M_total / (dt / args.steps)

    # dominant kernel (sigma/acquisition/arg-max): hipEvent pairs recorded on its stream inside the timed region
    k_ms, k_launches, k_cands = gp.read_profile() if not args.no_kernel_events else (1.0, 1, 1)
    k_avg_ms = k_ms / max(k_launches, 1)
    cand_per_launch = k_cands / max(k_launches, 1)
    flop_per_cand = float(N) * N + 2.0 * N           # triangular product N^2 + |v|^2 2N  
    achieved = flop_per_cand * cand_per_launch / (k_avg_ms * 1e-3) / 1e12
    traffic = None
    pmc = os.path.join(REPO, "profiles", "pmc_sigma_acq.json")
    default_shape = (N, d, args.m_per_gpu, args.dtype, args.chunk) == (512, 8, 1 << 20, "f64", 0)
    if os.path.exists(pmc) and default_shape:  # the committed PMC pass was taken on the default workload
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:  # noqa: BLE001
            traffic = None
    peak = FP32_MFMA_PEAK_TFLOPS if f32 else FP64_MFMA_PEAK_TFLOPS
    roofline = dict(bound="mfma", achieved=round(achieved, 3), peak=peak, unit="TFLOP/s",
                    frac=round(achieved / peak, 4), traffic=traffic,
                    kernel="sigma_acq_f32_kernel" if f32 else "sigma_acq_kernel", launches=int(k_launches), avg_launch_ms=round(k_avg_ms, 4),
                    flop_per_candidate=flop_per_cand, candidates_per_launch=cand_per_launch,
                    event_stride=int(max(args.event_stride, 1)))  # launches of every k-th timed step are bracketed

    # second kernel of the path: K(X*,X) build (HBM-write bound when materialised): algorithmic bytes per candidate
    # = N*w written + d*w read (w = 8 for fp64), DESIGN.md section 4
    kstar_roofline = None
    if not f32 and not args.no_kernel_events:
        ks_ms, ks_launches, ks_cands = gp.read_profile_kstar()
        if ks_launches:
            ks_avg = ks_ms / ks_launches
            bytes_per_cand = 8.0 * (N + d)
            gbs = bytes_per_cand * (ks_cands / ks_launches) / (ks_avg * 1e-3) / 1e9
            kstar_roofline = dict(bound="hbm", achieved=round(gbs, 1), peak=8000.0, unit="GB/s",
                                  frac=round(gbs / 8000.0, 4), kernel="kstar_mu_kernel", launches=int(ks_launches),
                                  avg_launch_ms=round(ks_avg, 4), bytes_per_candidate=bytes_per_cand)

    # time of the scoring part alone (factorisation excluded), for the record
    fence()
    t1 = time.perf_counter()
    for _ in range(max(3, args.steps // 4)):
        res, _, _, _ = score_async()
        gp.read_result(res)
    torch.cuda.synchronize(dev)
    ms_score = (time.perf_counter() - t1) / max(3, args.steps // 4) * 1e3

    cfg_name = {(512, 8, "f64"): "configs[1]", (4096, 8, "f64"): "configs[2] (per-GPU shard)",
                (8192, 16, "f32"): "configs[3] (per-GPU shard)"}.get((N, d, args.dtype), "custom")
    out = None
    if rank == 0:
        out = {
            "metric": "candidate acquisitions/sec", "value": value, "unit": "candidates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"{cfg_name}: d={d}, N={N} Sobol observations, M=2^{int(np.log2(args.m_per_gpu))} "
                                    f"Sobol candidates per GPU, ARD-SE GP, LCB(explore=4) arg-max, "
                                    f"{'fp64 factorisation + fp32 scoring' if f32 else 'fp64'}; "
                                    f"step = factorise + score all candidates + reduce"),
                       "candidates_total": M_total, "parallelism": f"candidate-sharded x{world}"},
            "ms_per_step_scoring_only": ms_score,
            "value_excl_factorisation": (hi - lo) * world / (ms_score * 1e-3),
            "argmax_index": best[1], "roofline": roofline, "kstar_roofline": kstar_roofline,
        }
        if world == 1 and not args.no_cpu_baseline and not f32:
            ns = min(args.cpu_sample, hi - lo)
            cb, idx_cpu = cpu_baseline(X, y, Xs_local[:ns], ls)
            r = gp.score(Xsd[:ns], acquisition="lcb", explore=4.0)
            cb["argmax_match_on_sample"] = bool(r.best_idx == idx_cpu)
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
