#!/usr/bin/env python3
"""bench.py - candidate acquisitions/sec of the GP acquisition path on MI355X.

One "step" = one BO inner-loop pass over one batch of synthetic input, everything resident in HBM
when the timed region starts:
    factorise K(X,X)+jitter (kxx, Cholesky, U = L^-T, alpha)            [once per step]
    K(X*,X) + mu + sigma + acquisition + arg-max                         [every candidate of the rank]
    one all-gather of (best value, lowest index, NaN count) across ranks [N > 1 only]
Workload (default): the configuration BASELINE.json's metric is quoted on - d=8, N=4096 Sobol observations,
fp64, 2^21 Sobol candidates per GPU (= configs[2]'s per-GPU shard: 8 ranks score 2^24), LCB(explore=4), the
reference's acquisition (point_selector.py:197-207).  `--acq ei` / `--acq qei` time the acquisitions the north star
names; configs[1] (N=512, M=2^20), configs[3] (--dtype f32 --d 16 --n-obs 8192 --m-per-gpu 524288) and configs[4]
(--acq qei --n-obs 2048 --m-per-gpu 1048576) stay reachable by flags; the default run also reports, under "also": EI on
the same workload, the same workload through the int8-sliced variance screen, configs[1], and configs[0] (the reference's
own sizes, on the fixture the reference produced, with the oracle timed beside it).
For N>1 the candidate set grows with N (contiguous shards): weak scaling; the reported value is the whole-job
rate M_total / max-over-ranks step time.

Usage: python bench.py --gpus N --steps K --warmup W
    N>1 without WORLD_SIZE in the environment: this process starts the N ranks itself (child processes under
    torch.distributed.run on 127.0.0.1) and exits with their status; under torchrun (WORLD_SIZE set) it is one rank.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = fp32 vector peak
I8_MFMA_PEAK_TOPS = 5033.0  # v_mfma_i32_32x32x32_i8: 2048 int8 op/clk/SIMD x 1024 SIMDs x 2.4 GHz (2x the bf16 rate)
I8_SLICE_PRODUCTS = 20     # int8 slice products per fp64-equivalent multiply-add (csrc/ozaki.hip)
I8C_SLICE_PRODUCTS = 6     # the coarse screen: three digits per operand, diagonals a + b <= 2
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense fp64 matrix peak = fp64 vector peak (half the 157.3 TF fp32 rate
#                               listed in MI355X_MICROARCH.md; AMD data sheet value)
HBM_PEAK_GBS = 8000.0
VALU_F64_PEAK_TLANE = 33.0  # fp64 vector lane-instructions/s (x1e12) the chip sustains: tools/valu_f64_peak.hip (66 TFLOP/s FMA)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-obs", type=int, default=4096)
    ap.add_argument("--m-per-gpu", type=int, default=1 << 21)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--acq", choices=["lcb", "ei", "qei"], default="lcb",
                    help="lcb: the reference's acquisition; ei: closed-form Expected Improvement; qei: q=8 Monte-Carlo "
                         "qEI with 512 fixed base samples (BASELINE configs[4])")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target wall time of the cpu_baseline leg (the sample size is chosen from a short probe)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="fix the cpu_baseline sample instead (candidates)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra configs[1] / EI measurements after the timed run")
    ap.add_argument("--backend", default="nccl", help="process-group backend; gloo only to rehearse N>1 on one GPU")
    ap.add_argument("--all-on-device", type=int, default=-1,
                    help="rehearsal only: put every rank on this GPU index instead of LOCAL_RANK")
    ap.add_argument("--event-stride", type=int, default=1,
                    help="bracket the kernels with HIP events in every k-th timed step only (an event record costs a few "
                         "microseconds of idle GPU, which matters for sub-millisecond launches; 1 = every step)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="diagnostic: do not bracket the kernels with HIP events (no roofline in the output)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="rehearsal only: initialise the process group and run the exchange step even at N=1")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher check (no GPU): every rank joins the process group, rank 0 prints n_gpus / ranks_seen")
    ap.add_argument("--dtype", choices=["f64", "f32", "i8", "i8c", "f64b"], default="f64",
                    help="f32: fp64 factorisation, fp32 screening of all candidates + fp64 re-scoring of the survivors "
                         "(BASELINE configs[3] shape); i8: the same with the variance product from int8 slices on the "
                         "integer matrix cores (|dsigma| ~ 1e-10)")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` with no rendezvous in the environment: start the N ranks as CHILD processes (this
    process has not touched the GPU - nothing is exec'ed over an initialised runtime) and relay their status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(X, y, Xs, ls, acq, target_s, fixed_sample, f_best, variance_dtype="f64", winner=None, check=True):
    """The oracle's Cholesky route (NumPy/LAPACK, BLAS threads = host cores) on a bounded sample of the
    same workload; factorisation excluded (it is amortised over the 2^21 candidates of a real step).  The sample is
    sized from a 1024-candidate probe so that the leg takes about `target_s` seconds."""
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
    N, d = X.shape
    _, L, alpha = O.factorise(X, y, ls)

    vdt = np.float32 if variance_dtype == "f32" else np.float64   # config 4: the fp32 restatement

    def run(P, with_value=False):
        mu, sig = O.posterior_chol(X, y, P, ls, L=L, alpha=alpha, variance_dtype=vdt)
        a = O.lcb(mu, sig, 4) if acq == "lcb" else O.expected_improvement(mu, sig, f_best, 0.0)
        i = int(np.flatnonzero(a == a.max())[0])
        return (i, float(a[i])) if with_value else i

    if fixed_sample:
        ns = min(fixed_sample, len(Xs))
    else:
        t0 = time.perf_counter()
        run(Xs[:1024])
        probe = time.perf_counter() - t0
        ns = int(min(len(Xs), max(2048, 1024 * target_s / max(probe, 1e-6))))
        ns = 1 << (ns.bit_length() - 1)  # power of two at or below the target
    t0 = time.perf_counter()
    idx = run(Xs[:ns])
    dt = time.perf_counter() - t0
    if not check:   # the timed leg only (the `also` entries: their arg-max is checked by tests/test_gpu_fullsize.py)
        cpu_baseline.last_window = None
        return dict(value=ns / dt, unit="candidate acquisitions/s", cores=int(threads), kind="port",
                    sample=f"first {ns} of the candidates, N={N}, d={d}, posterior + {acq.upper()} + arg-max, {dt:.1f} s wall, "
                           f"factorisation excluded (oracle/gp_oracle.py posterior_chol"
                           f"{', variance product in fp32' if variance_dtype == 'f32' else ''})"), idx, ns
    if vdt is np.float32:   # what is TIMED is the fp32 restatement; what the GPU's decision is checked against is fp64
        vdt = np.float64
    # Untimed: the oracle's (index, value) on the sample and on a window around the arg-max the GPU reported, so that the
    # check below speaks about the winner and not only about the first candidates.
    idx, val = run(Xs[:ns], with_value=True)
    window = None
    if winner is not None:
        w0 = max(0, min(int(winner) - 2048, len(Xs) - 4096))
        w1 = min(len(Xs), w0 + 4096)
        if w1 > ns:  # otherwise the sample already holds the winner
            wi, wv = run(Xs[w0:w1], with_value=True)
            window = (w0, w1)
            if wv > val or (wv == val and w0 + wi < idx):
                idx, val = w0 + wi, wv
    cpu_baseline.last_window = window
    return dict(value=ns / dt, unit="candidate acquisitions/s", cores=int(threads), kind="port",
                sample=f"first {ns} of the rank's candidates, N={N}, d={d}, posterior + {acq.upper()} + arg-max, "
                       f"{dt:.1f} s wall, factorisation excluded (oracle/gp_oracle.py posterior_chol"
                       f"{', variance product in fp32; arg-max checked against the fp64 route' if variance_dtype == 'f32' else ''})"), idx, ns


def _pmc_entry(N, d, dtype, cands):
    """The committed rocprofv3 --pmc entry of this shape (profiles/pmc_sigma_acq.json) and whether the kernel sources have
    changed since it was collected (profiles/source_hash.py): (entry or None, stale True / False / None = no hash stored)."""
    try:
        shapes = json.load(open(os.path.join(REPO, "profiles", "pmc_sigma_acq.json")))
        e = shapes.get(f"N={N},d={d},dtype={dtype},candidates_per_launch={int(cands)}")
        if not e:
            return None, None
        sys.path.insert(0, os.path.join(REPO, "profiles"))
        from source_hash import kernel_source_hash

        stale = (e["kernel_source_hash"] != kernel_source_hash(dtype)) if "kernel_source_hash" in e else None
        return e, stale
    except Exception:  # noqa: BLE001
        return None, None


def kernel_rooflines(gp, N, d, dtype, qei=False, event_stride=1):
    """(roofline, kstar_roofline, qei_roofline) of the launches `gp`'s profile has recorded since its last reset: the
    dominant kernel (sigma / acquisition / arg-max) on the matrix-core peak of its arithmetic type, the K(X*,X) build on
    HBM (or, on the bound route, on the fp64 vector-issue ceiling), the qEI stage on HBM.  ALGORITHMIC work per candidate
     x candidates per launch / the average launch time, hipEvents on the kernels' own stream."""
    f32, i8c, bnd = dtype == "f32", dtype == "i8c", dtype == "f64b"
    i8 = dtype in ("i8", "i8c")
    k_ms, k_launches, k_cands = gp.read_profile()
    if not k_launches:
        return None, None, None
    k_avg_ms = k_ms / k_launches
    cand_per_launch = k_cands / k_launches
    flop_per_cand = float(N) * N + 2.0 * N           # triangular product N^2 + |v|^2 2N  
    if bnd:   # the first pass multiplies the first J columns of V only (the re-scoring launches are not bracketed)
        Jp = float((gp.last_screen or {}).get("prefix", N))
        flop_per_cand = Jp * Jp + 2.0 * Jp
    achieved = flop_per_cand * cand_per_launch / (k_avg_ms * 1e-3) / 1e12
    # HBM bytes of one launch from the committed rocprofv3 --pmc passes of this same command line (profiles/):
    # counters cannot be read from inside the run, so the figure is replayed for the shape it was collected on - and
    # flagged when the kernel sources are no longer the ones that were profiled
    entry, stale = _pmc_entry(N, d, dtype, cand_per_launch)
    traffic = entry["hbm_bytes_per_launch"] if entry else None
    traffic_src = (f"committed PMC pass {entry['source']} (FETCH_SIZE x2 + WRITE_SIZE), not this run" if entry else None)
    peak = FP32_MFMA_PEAK_TFLOPS if f32 else FP64_MFMA_PEAK_TFLOPS
    unit = "TFLOP/s"
    if i8:  # algorithmic work of this kernel: 20 (coarse: 6) int8 slice products per multiply-add of the triangular product
        flop_per_cand = (I8C_SLICE_PRODUCTS if i8c else I8_SLICE_PRODUCTS) * float(N) * N
        achieved = flop_per_cand * cand_per_launch / (k_avg_ms * 1e-3) / 1e12
        peak, unit = I8_MFMA_PEAK_TOPS, "TOP/s (int8)"
    roofline = dict(bound="mfma", achieved=round(achieved, 3), peak=peak, unit=unit,
                    frac=round(achieved / peak, 4), traffic=traffic, traffic_source=traffic_src, traffic_stale=stale,
                    kernel={"f32": "sigma_acq_f32_kernel", "i8": "sigma_i8_kernel", "i8c": "sigma_i8c_kernel"}.get(
                        dtype, "sigma_acq_kernel"),
                    launches=int(k_launches),
                    avg_launch_ms=round(k_avg_ms, 4), flop_per_candidate=flop_per_cand,
                    candidates_per_launch=cand_per_launch,
                    event_stride=int(max(event_stride, 1)))  # launches of every k-th timed step are bracketed
    # second kernel of the path: K(X*,X) build (HBM-write bound when materialised): algorithmic bytes per
    # candidate = N*w written + d*8 read (w = 8 for fp64, 4 for the fp32 screen), DESIGN.md section 4
    kstar_roofline = None
    ks_ms, ks_launches, ks_cands = gp.read_profile_kstar()
    if ks_launches:
        ks_avg = ks_ms / ks_launches
        bytes_per_cand = {"f32": 4.0, "i8": 5.0, "i8c": 3.0}.get(dtype, 8.0) * N + 8.0 * d
        if bnd:
            # The prefix-bound route's K(X*,X) interval = kstar_mu_mfma_kernel (the mean of all N observations: 10 fp64
            # VALU instructions per (candidate, observation) pair beside 2 fp64 MFMAs per 256 pairs) +
            # kstar_mu_kernel on the J stored rows (41 per pair).  It writes J rows, not N: the bound is the vector
            # fp64 issue rate (33 T lane-instructions/s measured by tools/valu_f64_peak.hip), not HBM.  The MFMAs do not
            # run beside fp64 VALU on gfx950 (32 of the ~72 cycles per pair and lane), so frac cannot reach 1.
            Jp = float((gp.last_screen or {}).get("prefix", N))
            lane_instr_per_cand = 10.0 * N + 41.0 * Jp   # (round 5: 10 per pair with the one-term exponential, 19 before)
            tli = lane_instr_per_cand * (ks_cands / ks_launches) / (ks_avg * 1e-3) / 1e12
            kstar_roofline = dict(bound="valu", achieved=round(tli, 2), peak=VALU_F64_PEAK_TLANE,
                                  unit="T lane-instructions/s (fp64 VALU)", frac=round(tli / VALU_F64_PEAK_TLANE, 4),
                                  kernel="kstar_mu_mfma_kernel + kstar_mu_kernel on the stored rows",
                                  launches=int(ks_launches), avg_launch_ms=round(ks_avg, 4),
                                  lane_instructions_per_candidate=lane_instr_per_cand,
                                  note="static instruction counts per pair (ISA of the two kernels); the 2 fp64 MFMAs per 256 "
                                       "pairs of the first kernel occupy the SIMD for 32 of ~72 cycles per pair and lane and "
                                       "do not overlap with fp64 VALU on gfx950")
        else:
            gbs = bytes_per_cand * (ks_cands / ks_launches) / (ks_avg * 1e-3) / 1e9
            kstar_roofline = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                                  frac=round(gbs / HBM_PEAK_GBS, 4),
                                  kernel={"f32": "kstar_mu_kernel<..., float>", "i8": "kstar_slices_kernel<..., 5>",
                                          "i8c": "kstar_slices_kernel<..., 3>"}.get(dtype, "kstar_mu_kernel"),
                                  launches=int(ks_launches), avg_launch_ms=round(ks_avg, 4),
                                  bytes_per_candidate=bytes_per_cand)
            # the other resource this kernel loads: vector issue.  SQ_INSTS_VALU of the committed PMC pass of this shape
            # (a static property of the kernel and the shape, replayed like roofline.traffic), over THIS run's launch time
            e2, stale2 = _pmc_entry(N, d, dtype, ks_cands / ks_launches)
            wi = (e2 or {}).get("kstar_valu_wave_instructions_per_launch")
            if wi:
                tli = wi * 64.0 / (ks_avg * 1e-3) / 1e12
                kstar_roofline["valu"] = dict(
                    lane_instructions_per_s_T=round(tli, 2), fp64_issue_peak_T=VALU_F64_PEAK_TLANE,
                    frac=round(tli / VALU_F64_PEAK_TLANE, 4), wave_instructions_per_launch=wi,
                    source=f"SQ_INSTS_VALU of the committed PMC pass {e2['source']}, not this run", stale=stale2,
                    note="the kernel issues vector instructions at this share of the measured fp64 issue ceiling WHILE "
                         "streaming its stores: two nearly saturated resources that do not overlap perfectly")
    qei_roofline = None
    if qei:
        # the qEI stage (qei_kernel: the batch's 8 x 8 Gram block summed from the partials the variance launch left - round 5:
        # V itself no longer leaves the variance kernel's registers - then the 8 x 8 Cholesky and S samples).  What it reads:
        # 64 doubles per partial and batch, 2 partials per batch (16 when the variance launch runs in 8 column groups)
        q_ms, q_launches, q_cands = gp.read_profile_qei()
        if q_launches:
            q_avg = q_ms / q_launches
            parts = 16 if (q_cands / q_launches >= 32768 and gp.Np >= 2048) else 2
            bpc = parts * 64.0 * 8.0 / 8.0
            gbs = bpc * (q_cands / q_launches) / (q_avg * 1e-3) / 1e9
            qei_roofline = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                                frac=round(gbs / HBM_PEAK_GBS, 4), kernel="qei_kernel", launches=int(q_launches),
                                avg_launch_ms=round(q_avg, 4), bytes_per_candidate=bpc,
                                note="round 5: the joint posterior's Gram blocks are formed on the matrix cores inside the "
                                     "variance launch (sigma_acq_kernel<0, true>); this stage reads 64 x partials doubles per "
                                     "batch instead of the 8 N bytes per candidate of V it used to (2.15 GB per launch at N = 2048) "
                                     "and is bound by its own arithmetic (36 + S x 44 flop per batch of 8), not by HBM")
    if kstar_roofline is not None:
        # north_star's second number inside the dict the driver parses: the K(X*,X) build's share of its own roofline(s)
        roofline["kstar"] = dict(kernel=kstar_roofline["kernel"], bound=kstar_roofline["bound"], frac=kstar_roofline["frac"],
                                 valu_frac=(kstar_roofline.get("valu") or {}).get("frac"),
                                 avg_launch_ms=kstar_roofline["avg_launch_ms"])
    return roofline, kstar_roofline, qei_roofline


def host_threads():
    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:  # noqa: BLE001
        threads = os.cpu_count() or 1
    try:
        threads = min(threads, len(os.sched_getaffinity(0)))
    except Exception:  # noqa: BLE001
        pass
    return int(threads)


def _pmc_ard_entry(N, d):
    """The committed counter pass of the likelihood grid at this shape (profiles/pmc_ard.json, profiles/collect_ard.sh) and
    whether ard.hip has changed since."""
    try:
        e = json.load(open(os.path.join(REPO, "profiles", "pmc_ard.json"))).get(f"N={N},d={d},cells=2500")
        if not e:
            return None, None
        sys.path.insert(0, os.path.join(REPO, "profiles"))
        from source_hash import kernel_source_hash

        return e, e.get("kernel_source_hash") != kernel_source_hash(e.get("hash_key", "ard"))
    except Exception:  # noqa: BLE001
        return None, None


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as "
                 f"{args.gpus} GPUs")

    import numpy as np
    import torch
    import torch.distributed as dist

    use_pg = world > 1 or args.force_process_group
    if args.rendezvous_only:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "ranks_seen": int(t.item()), "rendezvous_only": True}), flush=True)
        dist.destroy_process_group()
        return

    dev_index = local_rank if args.all_on_device < 0 else args.all_on_device
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))  # only reached for a forced one-rank group
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    ranks_seen = dist.get_world_size() if use_pg else 1

    from bayesian_optimisation_amd import DeviceGP
    from bayesian_optimisation_amd import distributed as D
    from bayesian_optimisation_amd.synthetic import ard_length_scales, rff_objective, sobol_points

    N, d = args.n_obs, args.d
    f32 = args.dtype == "f32"
    i8c = args.dtype == "i8c"
    i8 = args.dtype == "i8" or i8c
    bnd = args.dtype == "f64b"   # fp64 throughout, prefix-bound screen (branch and bound) in front of the fp64 decision
    qei = args.acq == "qei"
    if qei and (f32 or i8 or bnd or args.m_per_gpu % 8):
        sys.exit("bench.py: --acq qei needs fp64 and a multiple of 8 candidates per GPU")
    M_total = args.m_per_gpu * world
    lo, hi = D.shard_bounds(M_total, world, rank)
    ls = ard_length_scales(d)
    X = sobol_points(0, N, d)
    y = rff_objective(X, ls)
    f_best = float(np.min(y))
    Xs_local = sobol_points(N + lo, hi - lo, d)  # this rank's contiguous shard of the Sobol candidate stream
    kw = dict(chunk=args.chunk) if args.chunk else {}
    gp = DeviceGP(dev, **kw)
    Xd, yd, Xsd = gp._dev(X), gp._dev(y), gp._dev(Xs_local)  # inputs resident in HBM before timing
    Zd = None
    if qei:
        Zd = gp._dev(np.random.default_rng(7).standard_normal((512, 8)))  # fixed base samples
    events = not args.no_kernel_events
    if events:
        gp.enable_profile(8192)
    acq_kw = dict(acquisition="lcb", explore=4.0) if args.acq == "lcb" else dict(acquisition="ei", f_best=f_best, xi=0.0)

    def score_async(g=gp, P=Xsd, off=lo):
        if qei:
            return g.score_qei_async(P, Zd, f_best=f_best, xi=0.0, batch_offset=off // 8)
        if f32:
            return g.score_async_f32(P, idx_offset=off, **acq_kw)
        if bnd:
            return g.score_async_bound(P, idx_offset=off, **acq_kw)
        if i8c:
            return g.score_async_i8c(P, idx_offset=off, **acq_kw)
        if i8:
            return g.score_async_i8(P, idx_offset=off, **acq_kw)
        return g.score_async(P, idx_offset=off, **acq_kw)

    def step(sample_events=True, g=gp, A=Xd, b=yd, P=Xsd, off=lo):
        g.profile_active = sample_events
        g.factorise(A, b, ls, check=False, order="fps" if bnd else "arrival")   # (f64b: farthest-point order, timed)
        if f32:
            g.prepare_f32()
        if i8:
            g.prepare_i8()
        score_async(g, P, off)
        # the one exchange step: the 40-byte device record (result + factorisation info) is gathered over the ranks
        # and read back once (at N=1: just the read-back, which synchronises this rank)
        v, i, n, info = D.allreduce_status(g.status, force_collective=args.force_process_group)
        if info != 0:
            raise RuntimeError("Cholesky failed")
        return v, i, n

    def fence():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        best = step()
    if events:
        gp.reset_profile()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        best = step(i % max(args.event_stride, 1) == 0)
    fence()
    dt = time.perf_counter() - t0
    gp.profile_active = True
    per_rank_ms = [dt / args.steps * 1e3]
    if use_pg:
        # every rank's own time, gathered over the process group (RCCL: device tensors); the step time is the MAX
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        allt = torch.empty(dist.get_world_size(), dtype=torch.float64, device=tt.device)
        dist.all_gather_into_tensor(allt, tt)
        per_rank_ms = [float(v) / args.steps * 1e3 for v in allt.cpu()]
        dt = max(float(v) for v in allt.cpu())
    ms_step = dt / args.steps * 1e3
    value = M_total / (dt / args.steps)

    roofline = kstar_roofline = qei_roofline = None
    if events:
        roofline, kstar_roofline, qei_roofline = kernel_rooflines(gp, N, d, args.dtype, qei, args.event_stride)

    # time of the scoring part alone (factorisation excluded), for the record
    fence()
    reps = max(2, args.steps // 4)
    t1 = time.perf_counter()
    for _ in range(reps):
        score_async()
        gp.read_result(gp.status[:4])
    torch.cuda.synchronize(dev)
    ms_score = (time.perf_counter() - t1) / reps * 1e3

    def timed_steps(fn, reps, g):
        """1 untimed + `reps` timed calls of fn() (each ends with the read-back of the result record); the kernels of the
        timed calls are bracketed with events in g's profile."""
        g.profile_active = False
        fn()
        torch.cuda.synchronize(dev)
        g.reset_profile()
        g.profile_active = True
        t = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t) / reps * 1e3, out

    def fact_entry(g, A, b, n, reps=5):
        """The once-per-step part of a (factorised) surrogate by itself, on the fp64 matrix-core roofline (2 n^3 / 3 flop)."""
        for _ in range(2):
            g.factorise(A, b, g.ls_h, check=False)
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(reps):
            g.factorise(A, b, g.ls_h, check=False)
        torch.cuda.synchronize(dev)
        fms = (time.perf_counter() - t) / reps * 1e3
        ftf = 2.0 * float(n) ** 3 / 3.0 / (fms * 1e-3) / 1e12
        return dict(ms_per_call=fms, flop=2.0 * float(n) ** 3 / 3.0, achieved_tflops=round(ftf, 2), peak=FP64_MFMA_PEAK_TFLOPS,
                    frac=round(ftf / FP64_MFMA_PEAK_TFLOPS, 4), steps=reps)

    def also_config3():
        """BASELINE configs[3], per-GPU shard: d=16, N=8192, 2^19 candidates, fp64 factorisation + fp32 variance screen +
        fp64 decision (--dtype f32 --d 16 --n-obs 8192 --m-per-gpu 524288 times it as the main workload)."""
        n3, d3, m3 = 8192, 16, 1 << 19
        ls3 = ard_length_scales(d3)
        X3 = sobol_points(0, n3, d3)
        y3 = rff_objective(X3, ls3)
        g3 = DeviceGP(dev)
        g3.enable_profile(256)
        A3, b3, P3 = g3._dev(X3), g3._dev(y3), g3._dev(sobol_points(n3, m3, d3))

        def one():
            g3.factorise(A3, b3, ls3, check=False)
            g3.prepare_f32()
            g3.score_async_f32(P3, acquisition="lcb", explore=4.0)
            return D.allreduce_status(g3.status)

        ms, (v, i, n, info) = timed_steps(one, 3, g3)
        rf, krf, _ = kernel_rooflines(g3, n3, d3, "f32")
        scr = dict(g3.last_screen)
        r64 = g3.score(P3[:1 << 15], acquisition="lcb", explore=4.0)      # the fp64 kernels on a slice that ...
        rs = g3.score_f32(P3[:1 << 15], acquisition="lcb", explore=4.0)   # ... the screened route must agree with
        fz = fact_entry(g3, A3, b3, n3)
        del g3, A3, b3, P3
        torch.cuda.empty_cache()
        # CPU: the fp32 restatement of this config on a fixed sample of 4,096 candidates (the 8192 x 8192
        # factorisation on the host, untimed, is the long part of this leg)
        cb3, _, _ = cpu_baseline(X3, y3, sobol_points(n3, 4096, d3), ls3, "lcb", 2.0, 4096, float(np.min(y3)), "f32", check=False)
        return dict(workload="configs[3] (per-GPU shard): d=16, N=8192, M=2^19, fp64 factorisation + fp32 variance screen + "
                             "fp64 re-score of the survivors, LCB(explore=4)", value=m3 / (ms * 1e-3), unit="candidates/s",
                    ms_per_step=ms, steps=3, dtype="f32", argmax_index=i, nan_count=n, screen=scr,
                    slice_argmax_matches_fp64=bool(r64.best_idx == rs.best_idx), roofline=rf, kstar_roofline=krf,
                    factorisation=fz, cpu_baseline=cb3)

    def also_config4():
        """BASELINE configs[4], per-GPU shard: q=8 Monte-Carlo qEI (512 fixed base samples), d=8, N=2048, 2^20 candidates
        (--acq qei --n-obs 2048 --m-per-gpu 1048576 times it as the main workload)."""
        n4, m4 = 2048, 1 << 20
        X4 = sobol_points(0, n4, d)
        y4 = rff_objective(X4, ls)
        g4 = DeviceGP(dev)
        g4.enable_profile(256)
        A4, b4, P4 = g4._dev(X4), g4._dev(y4), g4._dev(sobol_points(n4, m4, d))
        Z4 = g4._dev(np.random.default_rng(7).standard_normal((512, 8)))
        fb4 = float(np.min(y4))

        def one():
            g4.factorise(A4, b4, ls, check=False)
            g4.score_qei_async(P4, Z4, f_best=fb4, xi=0.0)
            return D.allreduce_status(g4.status)

        ms, (v, i, n, info) = timed_steps(one, 3, g4)
        rf, krf, qrf = kernel_rooflines(g4, n4, d, "f64", qei=True)
        fz = fact_entry(g4, A4, b4, n4)
        # CPU: the oracle's qEI (oracle.qei_mc: joint posterior of a batch, 8 x 8 Cholesky, the same 512 base samples) on the
        # first 256 batches; its own factorisation timed separately and subtracted (once per step, as on the GPU side)
        from oracle import gp_oracle as O

        Pq = sobol_points(n4, 2048, d)
        Zq = np.random.default_rng(7).standard_normal((512, 8))
        t = time.perf_counter()
        O.factorise(X4, y4, ls)
        tf_ = time.perf_counter() - t
        t = time.perf_counter()
        q_cpu = O.qei_mc(X4, y4, Pq, ls, Zq, f_best=fb4, xi=0.0)
        tq = max(time.perf_counter() - t - tf_, 1e-9)
        rq = g4.score_qei(P4[:2048], Z4, f_best=fb4, xi=0.0, dense=True)
        q_gpu = rq.acq.cpu().numpy() if rq.acq is not None else None
        cb4 = dict(value=2048 / tq, unit="candidate acquisitions/s", cores=host_threads(), kind="port",
                   sample=f"first 256 batches (2,048 candidates), N={n4}, d={d}, oracle.qei_mc, {tq:.2f} s wall, factorisation "
                          f"({tf_:.2f} s) excluded",
                   max_abs_diff_gpu_vs_oracle_on_sample=(float(np.max(np.abs(q_gpu - q_cpu))) if q_gpu is not None else None))
        del g4, A4, b4, P4
        torch.cuda.empty_cache()
        return dict(workload="configs[4] (per-GPU shard): q=8 Monte-Carlo qEI, 512 fixed base samples, d=8, N=2048, M=2^20 "
                             "(2^17 batches), fp64", value=m4 / (ms * 1e-3), unit="candidates/s", ms_per_step=ms, steps=3,
                    dtype="f64", argmax_batch=i, nan_count=n, qei_value=v, roofline=rf, kstar_roofline=krf, qei_roofline=qrf,
                    factorisation=fz, cpu_baseline=cb4)

    def also_ard_grid():
        """SURVEY 8(f) rank 1, the step in front of the hot path inside update_surrogate (point_selector.py:104-163): 2,500
        likelihood cells by themselves - the reference's own case (d=2, N=32, 50 x 50 cells) and the sizes of the BASELINE
        surrogates - on the fp64 matrix-core roofline (N^3 / 3 flop per cell: one Cholesky), HIP events on the launch stream,
        with the oracle's restatement of the reference's formula (inv + det per cell) on a sample of cells beside it."""
        from oracle import gp_oracle as O

        a = np.linspace(0.05, 3.0, 50)
        grid2 = np.stack(np.meshgrid(a, a, indexing="ij"), -1).reshape(-1, 2)
        out = []
        ga = DeviceGP(dev)
        for (n_, d_) in ((32, 2), (64, 2), (176, 2), (512, 8), (1024, 8)):   # (32, 64: the wave-per-cell kernel; beyond: the fused kernel)
            if d_ == 2:
                cells = grid2
            else:   # a d-feature search's cells: two coordinates over the 50 x 50 grid, the others at geomspace(0.2, 2)
                cells = np.tile(np.geomspace(0.2, 2.0, d_), (2500, 1))
                cells[:, :2] = grid2
            Xa = sobol_points(0, n_, d_)
            ya = rff_objective(Xa, ard_length_scales(d_))
            Ad, bd, cd = ga._dev(Xa), ga._dev(ya), ga._dev(cells)
            res_ref = ga.nlml_grid(Ad, bd, cd)                      # warm-up (allocates the workspace) + the values
            res_ld = ga.nlml_grid(Ad, bd, cd, likelihood="logdet")
            reps = 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ga.nlml_grid_device(Ad, bd, cd)
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(reps):
                o_dev = ga.nlml_grid_device(Ad, bd, cd)
            e1.record()
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / reps
            flop = 2500.0 * float(n_) ** 3 / 3.0
            tf = flop / (ms * 1e-3) / 1e12
            fused = n_ > int(ga.lib.gpbo_nlml_grid_wave_max_n())
            ent, stale = _pmc_ard_entry(n_, d_)
            rf = dict(bound="mfma" if fused else "valu", achieved=round(tf, 3), peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                      frac=round(tf / FP64_MFMA_PEAK_TFLOPS, 4),
                      kernel="nlml_fused_kernel (+ nlml_prep_kernel)" if fused else "nlml_wave_kernel",
                      avg_launch_ms=round(ms, 4), flop_per_cell=float(n_) ** 3 / 3.0, cells_per_launch=2500,
                      traffic=(ent or {}).get("fabric_bytes_per_call"), traffic_stale=stale,
                      algorithmic_bytes=(ent or {}).get("algorithmic_bytes_per_call"),
                      traffic_source=(f"committed PMC pass {ent['source']} (FETCH_SIZE x2 + WRITE_SIZE), not this run" if ent else None))
            if not fused:
                # N^2 / 2 fused multiply-adds per lane-instruction + two v_readlane each + the kernel entries' exp: counted as
                # vector instructions against the fp64 issue rate (16 lanes per cycle and SIMD), the matrix cores are not used
                instr = 2500.0 * (3.0 * n_ * n_ / 2.0 + n_ * (3.0 * d_ + 21.0) + 30.0 * n_)
                rf["valu_wave_instructions_per_launch"] = instr
                rf["valu_issue_frac"] = round(instr * 4.0 / (ms * 1e-3) / (1024 * 2.4e9), 4)
                rf["note"] = ("one wave per cell, the cell's matrix in registers (lane = row), column steps unrolled with v_readlane "
                              "broadcasts: bound by fp64 vector issue; frac is N^3/3 against the MATRIX peak for comparison with the "
                              "larger sizes, valu_issue_frac the estimated wave instructions x 4 cycles against 1,024 SIMDs at 2.4 GHz")
            # CPU: the oracle's restatement of eval_log_marginal (inv + det per cell), all host cores, a sample of cells
            nc = 64 if n_ <= 176 else (16 if n_ <= 512 else 6)
            sel = np.linspace(0, 2499, nc).astype(int)
            t = time.perf_counter()
            cpu = O.nlml_cells(Xa, ya, cells[sel])
            cdt = time.perf_counter() - t
            cpu_ld = O.nlml_cells_logdet(Xa, ya, cells[sel])
            fin = np.isfinite(cpu)
            ok_ref = bool(np.allclose(res_ref[sel][fin], cpu[fin], rtol=1e-5, atol=1e-2)) if fin.any() else None
            out.append(dict(
                workload=f"d={d_}, N={n_}, 2,500 likelihood cells (50 x 50 over two length scales), likelihood='reference'",
                ms_per_call=ms, cells_per_s=2500.0 / (ms * 1e-3), steps=reps, roofline=rf,
                finite_cells_reference_mode=int(np.isfinite(res_ref).sum()), finite_cells_logdet_mode=int(np.isfinite(res_ld).sum()),
                logdet_mode_max_rel_err_vs_oracle_on_sample=float(np.max(np.abs(res_ld[sel] - cpu_ld) / np.abs(cpu_ld))),
                reference_mode_matches_oracle_where_finite=ok_ref,
                cpu_baseline=dict(value=nc / cdt, unit="cells/s", cores=host_threads(), kind="port",
                                  sample=f"{nc} of the 2,500 cells (evenly spaced), oracle.nlml_cells (np.linalg.inv + det per cell, "
                                         f"point_selector.py:116-120), {cdt:.2f} s wall")))
        del ga
        torch.cuda.empty_cache()
        return out

    def also_append():
        """one more observation in O(N^2) (gpbo_append_f64) against factorising the N + 1 observations."""
        na = 4096
        Xa = sobol_points(0, na + 1, d)
        ya = rff_objective(Xa, ls)
        g5 = DeviceGP(dev)
        A5, b5 = g5._dev(Xa), g5._dev(ya)
        g5.factorise(A5[:na], b5[:na], ls, check=False)
        g5.append(A5[na], float(ya[na]), check=False)
        ts = []
        for _ in range(5):
            g5.factorise(A5[:na], b5[:na], ls, check=False)
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            g5.append(A5[na], float(ya[na]), check=False)
            torch.cuda.synchronize(dev)
            ts.append((time.perf_counter() - t) * 1e3)
        t = time.perf_counter()
        for _ in range(5):
            g5.factorise(A5, b5, ls, check=False)
        torch.cuda.synchronize(dev)
        fms = (time.perf_counter() - t) / 5 * 1e3
        ref = DeviceGP(dev).factorise(A5, b5, ls, check=False)
        g5.factorise(A5[:na], b5[:na], ls, check=False)
        g5.append(A5[na], float(ya[na]), check=False)
        da = float((g5.alpha[: na + 1] - ref.alpha[: na + 1]).abs().max() / ref.alpha[: na + 1].abs().max())
        del g5, ref
        torch.cuda.empty_cache()
        return dict(workload=f"d={d}, N=4096 -> 4097: gpbo_append_f64 (column N of U, alpha recomputed) vs gpbo_factorise_f64 of all 4097",
                    append_ms=float(np.median(ts)), refactorise_ms=fms, steps=5, alpha_max_rel_diff_vs_refactorisation=da,
                    note="host wall clock around one call each (append ends with the status read-back)")

    def also_full_m():
        """All 2^24 candidates of configs[2] in ONE call on one GPU (what 8 ranks share out): per-candidate rate, index
        width, workspace and the 128 chunk launches at the full count.  Shard r of the candidate set = this run's 2^21
        Sobol points shifted by r x the golden-ratio vector, mod 1 (a Cranley-Patterson rotation: generating 2^24 Sobol
        points on the host would take longer than the run); shard 0 is the headline run's own candidate set."""
        shards = 8
        shift = np.modf(np.outer(np.arange(shards), np.modf((np.arange(1, d + 1) * 0.6180339887498949))[0]))[0]
        Xall = torch.cat([torch.remainder(Xsd + gp._dev(shift[r])[None, :], 1.0) for r in range(shards)], 0)
        mloc = Xsd.shape[0]
        gp.factorise(Xd, yd, ls, check=False)
        per = [gp.score(Xall[r * mloc:(r + 1) * mloc], idx_offset=r * mloc, **acq_kw) for r in range(shards)]
        vb, ib = max(((r.best_val, -r.best_idx) for r in per))
        gp.profile_active = False
        gp.score_async(Xall, idx_offset=0, **acq_kw)        # untimed warm-up (workspace of the full count)
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        gp.factorise(Xd, yd, ls, check=False)
        gp.score_async(Xall, idx_offset=0, **acq_kw)
        v, i, n, info = D.allreduce_status(gp.status)
        ms = (time.perf_counter() - t) * 1e3
        gp.profile_active = True
        mall = int(Xall.shape[0])
        del Xall
        torch.cuda.empty_cache()
        return dict(workload=f"d={d}, N={N}, M=2^24 candidates in one call on one GPU, fp64, {args.acq.upper()}",
                    value=mall / (ms * 1e-3), unit="candidates/s", ms_per_step=ms, steps=1, candidates=mall,
                    argmax_index=i, nan_count=n, equals_reduction_of_8_shard_calls=bool(i == -ib and v == vb),
                    shard0_is_the_headline_run=bool(per[0].best_idx == best[1] - lo),
                    per_candidate_rate_vs_headline=round((mall / (ms * 1e-3)) / value, 4))

    def also():
        """More numbers from the same build on the same box (N=1 only): every BASELINE config with its own roofline, EI
        and the screened / bounded routes on the default workload - a few steps each, outside the timed region above."""
        res = {}
        reps2 = 3
        if args.acq == "lcb" and args.dtype == "f64":
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for _ in range(reps2):
                gp.factorise(Xd, yd, ls, check=False)
                gp.score_async(Xsd, acquisition="ei", f_best=f_best, xi=0.0, idx_offset=lo)
                v, i, n, info = D.allreduce_status(gp.status)
            ms = (time.perf_counter() - t) / reps2 * 1e3
            res["ei_same_workload"] = dict(value=(hi - lo) / (ms * 1e-3), unit="candidates/s", ms_per_step=ms,
                                           argmax_index=i, steps=reps2)
        if args.dtype == "f64" and N <= 16384 and not qei:
            # the same workload with the variance product from int8 slices on the integer matrix cores (csrc/ozaki.hip:
            # |dsigma| ~ 1e-10, means and the selected point still from the fp64 kernels)
            gp.prepare_i8()
            kw8 = dict(idx_offset=lo, **acq_kw)
            gp.score_async_i8(Xsd, **kw8)
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for _ in range(reps2):
                gp.factorise(Xd, yd, ls, check=False)
                gp.prepare_i8()
                gp.score_async_i8(Xsd, **kw8)
                v, i, n, info = D.allreduce_status(gp.status)
            ms = (time.perf_counter() - t) / reps2 * 1e3
            res["int8_sliced_same_workload"] = dict(
                value=(hi - lo) / (ms * 1e-3), unit="candidates/s", ms_per_step=ms, argmax_index=i,
                argmax_matches_fp64=bool(i == best[1]), steps=reps2, screen=gp.last_screen,
                note="variance product = 20 exact int8 slice products on v_mfma_i32_32x32x32_i8 (|dsigma| ~ 1e-10 against "
                     "the fp64 kernels); means and the selected point are the fp64 kernels'; --dtype i8 times it as the "
                     "main workload")
            # ... and with the coarse screen in front of the same fp64 decision (three digits per operand, six products)
            gp.score_async_i8c(Xsd, **kw8)
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for _ in range(reps2):
                gp.factorise(Xd, yd, ls, check=False)
                gp.prepare_i8()
                gp.score_async_i8c(Xsd, **kw8)
                v, i, n, info = D.allreduce_status(gp.status)
            ms = (time.perf_counter() - t) / reps2 * 1e3
            res["int8_coarse_screen_same_workload"] = dict(
                value=(hi - lo) / (ms * 1e-3), unit="candidates/s", ms_per_step=ms, argmax_index=i,
                argmax_matches_fp64=bool(i == best[1]), steps=reps2, screen=gp.last_screen,
                note="screen = 6 int8 slice products (three leading digits of K* and U, |dsigma^2| ~ 2e-4, tolerance checked "
                     "per call); every candidate whose interval reaches the best lower bound is re-scored by the fp64 "
                     "kernels, which decide: same selected point; --dtype i8c times it as the main workload")
        if args.dtype == "f64" and not qei:
            # ... and with NO approximation at all: branch and bound on the variance reduction of the first N/16 observations
            kwb = dict(idx_offset=lo, **acq_kw)
            gp.factorise(Xd, yd, ls, check=False, order="fps")   # the route's factorisation: farthest-point order
            gp.score_async_bound(Xsd, **kwb)
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for _ in range(reps2):
                gp.factorise(Xd, yd, ls, check=False, order="fps")
                gp.score_async_bound(Xsd, **kwb)
                v, i, n, info = D.allreduce_status(gp.status)
            ms = (time.perf_counter() - t) / reps2 * 1e3
            # how many candidates the bound fails to dispose of, by acquisition (one untimed call each): the route's speed
            # depends on the data and on the exploration weight - LCB with a large weight is the hard case
            surv = {}
            for name, kwx in (("lcb_explore_1", dict(acquisition="lcb", explore=1.0)),
                              ("lcb_explore_4", dict(acquisition="lcb", explore=4.0)),
                              ("lcb_explore_10", dict(acquisition="lcb", explore=10.0)),
                              ("ei", dict(acquisition="ei", f_best=f_best, xi=0.0))):
                rb = gp.score_bound(Xsd, idx_offset=lo, **kwx)
                tb = time.perf_counter()
                rb = gp.score_bound(Xsd, idx_offset=lo, **kwx)   # (synchronous: reads the result back)
                tb = (time.perf_counter() - tb) * 1e3
                ls_ = gp.last_screen or {}
                r6 = gp.score(Xsd, idx_offset=lo, **kwx)
                surv[name] = dict(survivors_first_level=ls_.get("survivors"), rescored_in_fp64=ls_.get("rescored"),
                                  fallback=ls_.get("fallback"), ms_scoring=round(tb, 2),
                                  same_point_as_plain_pass=bool(rb.best_idx == r6.best_idx))
            gp.score_async_bound(Xsd, **kwb)
            res["prefix_bound_screen_same_workload"] = dict(
                value=(hi - lo) / (ms * 1e-3), unit="candidates disposed/s", ms_per_step=ms, argmax_index=i,
                argmax_matches_fp64=bool(i == best[1]), steps=reps2, screen=gp.last_screen, survivors_by_acquisition=surv,
                note="fp64 throughout, exact: the mean of every candidate, an UPPER bound of its acquisition from |v|^2 over the "
                     "first N/16 components (1/256 of the variance product; N/4 for the survivors), the fp64 kernels on every "
                     "candidate whose bound reaches the best exact value seen; pruned candidates provably cannot be the "
                     "maximum nor tie with it; "
                     "--dtype f64b times it as the main workload")
            if args.acq == "lcb":   # the north star counts EI evaluations: the same route with Expected Improvement
                kwe = dict(idx_offset=lo, acquisition="ei", f_best=f_best, xi=0.0)
                gp.factorise(Xd, yd, ls, check=False)
                gp.score_async(Xsd, **kwe)
                v64, i64, n64, info = D.allreduce_status(gp.status)
                gp.factorise(Xd, yd, ls, check=False, order="fps")
                gp.score_async_bound(Xsd, **kwe)
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
                for _ in range(reps2):
                    gp.factorise(Xd, yd, ls, check=False, order="fps")
                    gp.score_async_bound(Xsd, **kwe)
                    v, i, n, info = D.allreduce_status(gp.status)
                ms = (time.perf_counter() - t) / reps2 * 1e3
                res["prefix_bound_screen_ei_same_workload"] = dict(
                    value=(hi - lo) / (ms * 1e-3), unit="candidates disposed/s", ms_per_step=ms, argmax_index=i,
                    argmax_matches_fp64=bool(i == i64), steps=reps2, screen=gp.last_screen,
                    note="EI is EVALUATED for the candidates the fp64 kernels re-score (screen.rescored) and bounded from above "
                         "for all the others - a rate of candidates disposed of, exactly, not of EI evaluations")
        g1 = os.path.join(REPO, "tests", "golden", "g1_m32.npz")
        if (N, d, args.dtype, args.acq) == (4096, 8, "f64", "lcb") and os.path.exists(g1):
            # BASELINE configs[0] (d=2, N=32, M=32x32 grid, 50x50 ARD search - the sizes the reference's DAG runs): the
            # drop-in class on the committed fixture the REFERENCE produced (tests/golden/make_golden.py), host arrays in and
            # out, against the oracle on the host cores and the reference's own selected point
            from bayesian_optimisation_amd import PointSelector
            from oracle import gp_oracle as O

            g = dict(np.load(g1))
            fd = [int(v) for v in g["feature_domain"]]

            def dropin():
                ps = PointSelector(device=dev)
                ps.name, ps.iteration = "T", 0
                ps.measured_pts, ps.measured_vals = g["X"], g["y"]
                ps.feature_domain, ps.predicted_pts, ps.length_scales = fd, g["Xs"], g["length_scales"]
                ps.update_surrogate()
                return ps.lower_confidence_bound()

            dropin(); dropin()
            ts = []
            for _ in range(15):
                t = time.perf_counter()
                idx = dropin()
                ts.append(time.perf_counter() - t)
            ms = float(np.median(ts)) * 1e3
            t = time.perf_counter()
            o = O.select_next(g["X"], g["y"], g["Xs"], fd, length_scales=g["length_scales"])
            cpu_ms = (time.perf_counter() - t) * 1e3
            res["configs[0]"] = dict(workload="d=2, N=32, M=32x32, 50x50 ARD grid + posterior + LCB through the drop-in class",
                                     value=len(g["Xs"]) / (ms * 1e-3), unit="candidates/s", ms_per_step=ms, steps=15,
                                     index=[int(v) for v in idx], reference_index=[int(v) for v in g["index"]],
                                     index_matches_reference=bool(np.array_equal(idx, g["index"])),
                                     cpu_port_ms=cpu_ms, cpu_port_index_matches=bool(np.array_equal(o["index"], g["index"])),
                                     note="the reference itself took 587 ms on this shape in the survey container (BASELINE.md 2)")
        if (N, d, args.dtype, args.acq) == (4096, 8, "f64", "lcb"):
            n2, m2 = 512, 1 << 20
            X2 = sobol_points(0, n2, d)
            y2 = rff_objective(X2, ls)
            g2 = DeviceGP(dev)
            A2, b2, P2 = g2._dev(X2), g2._dev(y2), g2._dev(sobol_points(n2, m2, d))
            for _ in range(2):
                step(False, g2, A2, b2, P2, 0)
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for _ in range(20):
                v, i, n = step(False, g2, A2, b2, P2, 0)
            ms = (time.perf_counter() - t) / 20 * 1e3
            cb1, _, _ = cpu_baseline(X2, y2, sobol_points(n2, 1 << 16, d), ls, "lcb", 2.0, 0, float(np.min(y2)), check=False)
            res["configs[1]"] = dict(workload="d=8, N=512, M=2^20, fp64, LCB(explore=4)", value=m2 / (ms * 1e-3),
                                     unit="candidates/s", ms_per_step=ms, argmax_index=i, steps=20, cpu_baseline=cb1)
        if (N, d, args.dtype, args.acq) == (4096, 8, "f64", "lcb"):
            res["configs[3]"] = also_config3()
            res["configs[4]"] = also_config4()
            res["ard_grid"] = also_ard_grid()
            res["append"] = also_append()
            res["configs[2]_all_2^24_candidates_on_one_gpu"] = also_full_m()
        if not qei:
            # the once-per-step part by itself: K(X,X) + fused Cholesky / inverse factor + alpha (gpbo_factorise_f64), on the
            # matrix-core roofline that bounds it (2 N^3 / 3 flop: Cholesky + triangular inverse)
            for _ in range(3):
                gp.factorise(Xd, yd, ls, check=False)
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            for _ in range(20):
                gp.factorise(Xd, yd, ls, check=False)
            torch.cuda.synchronize(dev)
            fms = (time.perf_counter() - t) / 20 * 1e3
            ftf = 2.0 * float(N) ** 3 / 3.0 / (fms * 1e-3) / 1e12
            res["factorisation"] = dict(ms_per_call=fms, flop=2.0 * float(N) ** 3 / 3.0, achieved_tflops=round(ftf, 2),
                                        peak=FP64_MFMA_PEAK_TFLOPS, frac=round(ftf / FP64_MFMA_PEAK_TFLOPS, 4),
                                        kernel="cholinv_kernel (+ kxx_kernel, transpose_w_kernel, utv / uv)", steps=20,
                                        note="latency-bound below N ~ 4096: 2 dependent launches per 128 rows (DESIGN.md 4e)")
        return res

    cfg_name = {(512, 8, "f64", "lcb"): "configs[1]", (4096, 8, "f64", "lcb"): "configs[2] (per-GPU shard)",
                (4096, 8, "f64", "ei"): "configs[2] (per-GPU shard), EI",
                (8192, 16, "f32", "lcb"): "configs[3] (per-GPU shard)",
                (4096, 8, "i8", "lcb"): "configs[2] (per-GPU shard), int8-sliced variance screen",
                (4096, 8, "i8c", "lcb"): "configs[2] (per-GPU shard), coarse int8 variance screen",
                (4096, 8, "f64b", "lcb"): "configs[2] (per-GPU shard), prefix-bound screen",
                (4096, 8, "f64b", "ei"): "configs[2] (per-GPU shard), EI, prefix-bound screen",
                (2048, 8, "f64", "qei"): "configs[4] (per-GPU shard)"}.get((N, d, args.dtype, args.acq), "custom")
    acq_txt = {"lcb": "LCB(explore=4) arg-max", "ei": "Expected Improvement (f_best=min y, xi=0) arg-max",
               "qei": "q=8 Monte-Carlo qEI (512 fixed base samples) arg-max over batches"}[args.acq]
    if rank == 0:
        lg = int(np.log2(args.m_per_gpu))
        mtxt = f"2^{lg}" if args.m_per_gpu == 1 << lg else str(args.m_per_gpu)
        out = {
            "metric": "candidate acquisitions/sec", "value": value, "unit": "candidates/s", "n_gpus": world,
            "ranks_seen": ranks_seen,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64" if bnd else args.dtype, "data": "synthetic",
            "config": {"workload": (f"{cfg_name}: d={d}, N={N} Sobol observations, M={mtxt} "
                                    f"Sobol candidates per GPU, ARD-SE GP, {acq_txt}, "
                                    f"{ {'f32': 'fp64 factorisation + fp32 screen + fp64 re-score of the survivors', 'i8': 'fp64 factorisation and means + int8-sliced variance screen + fp64 re-score of the survivors', 'i8c': 'fp64 factorisation and means + coarse int8 variance screen (three digits per operand, six slice products) + fp64 re-score of the survivors', 'f64b': 'fp64 throughout: mean of every candidate, UPPER bound of its acquisition from the variance reduction of the first N/16 observations of the farthest-point-ordered factorisation (N/4 for the survivors), fp64 re-score of every candidate whose bound reaches the best exact value (branch and bound, exact)'}.get(args.dtype, 'fp64') }; "
                                    f"step = factorise + score all candidates + reduce"),
                       "candidates_total": M_total, "parallelism": f"candidate-sharded x{world}"},
            "ms_per_step_scoring_only": ms_score,
            "value_excl_factorisation": (hi - lo) * world / (ms_score * 1e-3),
            "argmax_index": best[1], "roofline": roofline, "kstar_roofline": kstar_roofline,
            # one entry per rank (len = ranks_seen): the slowest one is ms_per_step
            "ms_per_step_by_rank": {"min": min(per_rank_ms), "max": max(per_rank_ms), "ranks": len(per_rank_ms)},
            "multi_gpu_note": ("value is measured on the ranks of THIS run only (n_gpus / ranks_seen); any 8-GPU figure quoted "
                               "elsewhere for a 1-GPU run is that number times 8 - an extrapolation, not a measurement"),
        }
        if qei_roofline is not None:
            out["qei_roofline"] = qei_roofline
        if world == 1 and not args.no_cpu_baseline and not qei:
            cb, idx_cpu, ns = cpu_baseline(X, y, Xs_local, ls, args.acq, args.cpu_seconds, args.cpu_sample, f_best,
                                           "f32" if f32 else "f64", winner=best[1] - lo)
            score_fn = {"f32": gp.score_f32, "i8": gp.score_i8, "i8c": gp.score_i8c, "f64b": gp.score_bound}.get(
                args.dtype, gp.score)
            r = score_fn(Xsd[:ns], **acq_kw)
            gpu_idx, gpu_val = r.best_idx, r.best_val
            win = cpu_baseline.last_window
            if win is not None:  # the same union on the GPU: the sample and the window around the reported arg-max
                rw = score_fn(Xsd[win[0]:win[1]], idx_offset=win[0], **acq_kw)
                if rw.best_val > gpu_val or (rw.best_val == gpu_val and rw.best_idx < gpu_idx):
                    gpu_idx, gpu_val = rw.best_idx, rw.best_val
            cb["argmax_match_on_sample"] = bool(gpu_idx == idx_cpu)
            cb["sample_contains_reported_argmax"] = bool(best[1] - lo < ns or (win is not None and win[0] <= best[1] - lo < win[1]))
            cb["reported_argmax_is_the_samples"] = bool(gpu_idx == best[1] - lo)
            if win is not None:
                cb["sample"] += f"; arg-max check (untimed): that sample plus candidates [{win[0]}, {win[1]}) around the reported arg-max"
            out["cpu_baseline"] = cb
        if f32 or i8 or bnd:
            out["screen"] = gp.last_screen   # survivors of the screen, tolerance / threshold and its check, fallback flag
        if world == 1 and not args.no_also:
            out["also"] = also()
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
