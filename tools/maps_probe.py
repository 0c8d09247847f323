"""Which runtime libraries are mapped, in both load orders of libgpbo.so and PyTorch (one process per order).

Background: a process that used the host-pointer binding first (libgpbo.so initialises HIP) and
brought PyTorch's GPU context up afterwards dead-locked twice in about 40 runs in round 1; the other order never did.
This prints every mapped library of the ROCm stack after each step, so that a duplicated runtime (two libamdhip64 /
libhsa-runtime64 / libamd_comgr / librocprofiler-register from different directories) would show.
usage: python tools/maps_probe.py lib_first|torch_first [cuda]     ("cuda": also bring PyTorch's GPU context up)"""
import faulthandler
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PAT = re.compile(r"(amdhip64|hsa-|amd_comgr|rocprofiler|roctx|roctracer|hiprtc|rocm-core|rocm_smi|amd_smi|libdrm|aqlprofile|rccl|hipblas|rocblas)")


def libs(tag):
    seen = sorted({l.split()[-1] for l in open("/proc/self/maps") if PAT.search(l.split()[-1] if l.split() else "")})
    base = {}
    for s in seen:
        base.setdefault(re.sub(r"\.so.*", "", os.path.basename(s)), []).append(s)
    dup = {k: v for k, v in base.items() if len({os.path.dirname(p) for p in v}) > 1}
    print(f"== {tag}: {len(seen)} libraries, duplicated across directories: {dup if dup else 'none'}")
    for s in seen:
        print("   ", s)
    sys.stdout.flush()


def host_call():
    import numpy as np

    from bayesian_optimisation_amd import host_binding as H

    rng = np.random.default_rng(0)
    X, y, Xs = rng.uniform(0, 1, (10, 2)), rng.standard_normal(10), rng.uniform(0, 1, (100, 2))
    return H.select_next(X, y, [0.3, 0.3], Xs)["best_idx"]


faulthandler.dump_traceback_later(45, exit=True)  # a dead-lock leaves its Python stack on stderr
order = sys.argv[1]
cuda = len(sys.argv) > 2
if order == "lib_first":
    print("host call ->", host_call())
    libs("after the host-pointer call (HIP initialised by libgpbo), PyTorch not imported")
    import torch

    libs("after import torch")
    if cuda:
        print("torch.cuda.is_available():", torch.cuda.is_available())
        t = torch.ones(4, device="cuda")
        torch.cuda.synchronize()
        libs("after PyTorch's GPU context came up")
else:
    import torch

    libs("after import torch")
    print("host call ->", host_call())
    libs("after the host-pointer call")
    if cuda:
        t = torch.ones(4, device="cuda")
        torch.cuda.synchronize()
        libs("after PyTorch's GPU context came up")
print("done")
