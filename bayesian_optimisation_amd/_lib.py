"""ctypes binding of libgpbo.so (the C ABI declared in include/gpbo.h).

The library is built in-tree by `bayesian_optimisation_amd/csrc/build.sh` (or `__graft_entry__.build()`).
There is NO fallback: if the shared object is missing or a symbol is absent, loading raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPBO_LIB=/path/to/variant.so: A/B and timing-only builds (tools/ab*.sh, tools/build_variant.sh) are loaded from where
# they were built; the installed library is never overwritten by a tool.
LIB_PATH = os.environ.get("GPBO_LIB") or os.path.join(_HERE, "libgpbo.so")

GPBO_OK = 0
ACQ_LCB = 0
ACQ_EI = 1
NPAD = 128
CHUNK_GRANULE = 512
MAX_D = 16          # every route
MAX_D_ANY = 1024    # fp64 route (factorise, score): any d up to this, slow path beyond MAX_D
I8_MAX_N = 16384

_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_f64 = C.c_double

# name -> (restype, argtypes); mirrors include/gpbo.h one to one
SIGNATURES = {
    "gpbo_version": (C.c_int, []),
    "gpbo_strerror": (C.c_char_p, [C.c_int]),
    "gpbo_padded_n": (_i64, [_i64]),
    "gpbo_kxx_f64": (C.c_int, [_p, _i64, _i32, _p, _f64, _f64, _p, _i64, _p]),
    "gpbo_potrf_f64": (C.c_int, [_p, _i64, _p, _p, _p]),
    "gpbo_trtri_f64": (C.c_int, [_p, _p, _i64, _p, _p, _p]),
    "gpbo_cholinv_f64": (C.c_int, [_p, _i64, _i64, _p, _p, _p]),
    "gpbo_cholinv_plan": (C.c_int, [_i64, _p, _p, _p, _p, _p]),
    "gpbo_cholinv_tiles_f64": (C.c_int, [_p, _i64, _i64, _p, _i32, _p, _i64, _i32, _i32, _p]),
    "gpbo_alpha_f64": (C.c_int, [_p, _p, _i64, _i64, _p, _p, _p]),
    "gpbo_factorise_workspace_bytes": (_i64, [_i64]),
    "gpbo_factorise_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _f64, _f64, _i64, _p, _p, _p, _p, _p, _i64, _p]),
    "gpbo_select_next_host_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _f64, _f64, _p, _i64, _i32, _f64, _f64, _f64, _i64,
                                            _p, _p, _p, _p, _p, _p]),
    "gpbo_select_qei_host_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _f64, _f64, _p, _i64, _f64, _f64, _p, _i32, _i64, _p,
                                           _p, _p]),
    "gpbo_nlml_grid_host_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _f64, _p]),
    "gpbo_nlml_grid_logdet_host_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _f64, _p]),
    "gpbo_append_workspace_bytes": (_i64, [_i64]),
    "gpbo_append_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _f64, _f64, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "gpbo_scale_points_f64": (C.c_int, [_p, _i64, _i64, _i32, _p, _p, _p]),
    "gpbo_kstar_mu_f64": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _f64, _i64, _p, _i64, _p, _p]),
    "gpbo_posterior_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "gpbo_posterior_acq_f64": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _f64, _i32, _f64, _f64, _f64,
                                         _i64, _i64, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "gpbo_profile_create": (C.c_int, [_i32, C.POINTER(_p)]),
    "gpbo_profile_reset": (None, [_p]),
    "gpbo_profile_read": (C.c_int, [_p, C.POINTER(_f64), C.POINTER(_i64), C.POINTER(_i64)]),
    "gpbo_profile_read_kstar": (C.c_int, [_p, C.POINTER(_f64), C.POINTER(_i64), C.POINTER(_i64)]),
    "gpbo_profile_read_qei": (C.c_int, [_p, C.POINTER(_f64), C.POINTER(_i64), C.POINTER(_i64)]),
    "gpbo_profile_destroy": (None, [_p]),
    "gpbo_qei_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "gpbo_posterior_qei_f64": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _f64, _f64, _f64, _p, _i32, _i64,
                                         _i64, _p, _p, _p, _i64, _p, _p]),
    "gpbo_padded_n_f32": (_i64, [_i64]),
    "gpbo_prepare_f32": (C.c_int, [_p, _p, _i64, _p, _p, _i64, _p]),
    "gpbo_posterior_workspace_bytes_f32": (_i64, [_i64, _i64, _i64]),
    "gpbo_posterior_acq_f32": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _f64, _i32, _f64, _f64, _f64,
                                         _i64, _i64, _p, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "gpbo_prepare_i8_bytes": (_i64, [_i64]),
    "gpbo_prepare_i8": (C.c_int, [_p, _i64, _p, _i64, _p]),
    "gpbo_posterior_workspace_bytes_i8": (_i64, [_i64, _i64, _i64]),
    "gpbo_posterior_acq_i8": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _f64, _i32, _f64, _f64, _i64, _i64,
                                        _p, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "gpbo_posterior_acq_i8c": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _f64, _i32, _f64, _f64, _i64, _i64,
                                         _p, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "gpbo_posterior_prefix_f64": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _f64, _i32, _f64, _f64, _i64, _i64,
                                            _i64, _p, _p, _p, _p, _p, _i64, _p, _p]),
    "gpbo_bound_select_f64": (C.c_int, [_p, _i64, _p, _p, _i64, _i64, _i32, _p, _p, _p, _f64, _i32, _f64, _f64, _i64,
                                        _i64, _i64, _i64, _i64, _p, _p, _p, _i64, _p]),
    "gpbo_fps_order_workspace_bytes": (_i64, [_i64]),
    "gpbo_fps_order_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _p, _p, _p, _p, _i64, _p]),
    "gpbo_fps_order_status": (C.c_int, [_p, _i64, _p, _p]),
    "gpbo_rescore_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "gpbo_rescore_f64": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _i64, _i32, _p, _p, _p, _f64, _i32, _f64, _f64, _i64,
                                   _f64, _i64, _i64, _i64, _p, _p, _p, _i64, _p]),
    "gpbo_acq_workspace_bytes": (_i64, []),
    "gpbo_acq_argmax_f64": (C.c_int, [_p, _p, _i64, _i32, _f64, _f64, _i64, _p, _p, _p, _i64, _p]),
    "gpbo_nlml_grid_max_n": (C.c_int, []),
    "gpbo_nlml_grid_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _f64, _p, _p]),
    "gpbo_nlml_grid_wave_max_n": (C.c_int, []),
    "gpbo_nlml_grid_wave_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _f64, _p, _p]),
    "gpbo_nlml_grid_wave_logdet_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _f64, _p, _p]),
    "gpbo_nlml_grid_batched_workspace_bytes": (_i64, [_i64, _i64]),
    "gpbo_nlml_grid_batched_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _f64, _p, _p, _i64, _p]),
    "gpbo_nlml_grid_batched_logdet_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _f64, _p, _p, _i64, _p]),
    "gpbo_nlml_cell_f64": (C.c_int, [_p, _p, _p, _i64, _i64, _p, _p, _p]),
    "gpbo_gemm_f64": (C.c_int, [_i32, _i64, _i64, _i64, _f64, _p, _i64, _i64, _p, _i64, _i64, _f64, _p, _i64, _i64,
                                _i32, _i32, _p]),
}

_lib = None
hip_used_before_pytorch = False  # a host-pointer entry point initialised HIP while PyTorch was not imported


class ScreenStats(C.Structure):
    """gpbo_screen_stats (include/gpbo.h)."""
    _fields_ = [("survivors", _i64), ("rescored", _i64), ("rounds", _i32), ("fallback", _i32), ("tau", _f64),
                ("err_max", _f64)]


class GpboError(RuntimeError):
    pass


def _bind_to_pytorch_hip_runtime():
    """Two HIP runtimes in one process do not share the device: whichever initialises second sees no GPU.
    PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's, which libgpbo.so would otherwise pull in).
    If PyTorch is installed but not imported yet (the NumPy-only host binding), load ITS runtime first, so that a
    later `import torch` in the same process finds the runtime it expects already in place.  Nothing is imported."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return  # its runtime is loaded; libgpbo.so's NEEDED libamdhip64.so.7 resolves to it
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    # The whole ROCm stack PyTorch bundles, in dependency order, so that a later `import torch` finds every one of its
    # runtime libraries already in place and from ITS directory (tools/maps_probe.py shows what is mapped in each
    # load order; with only libamdhip64 preloaded the system ROCm's libhsa-amd-aqlprofile64 ended up inside PyTorch's
    # HSA runtime in the library-first order and in no other).
    for name in ("librocm-core.so", "libdrm.so", "libdrm_amdgpu.so", "libnuma.so", "libelf.so", "librocprofiler-register.so",
                 "libhsa-runtime64.so", "libamd_comgr.so", "libamdhip64.so", "libhiprtc.so", "libroctx64.so",
                 "libroctracer64.so"):
        rt = os.path.join(libdir, name)
        if os.path.exists(rt):
            try:
                C.CDLL(rt, mode=C.RTLD_GLOBAL)
            except OSError:
                pass  # fall back to whatever the dynamic linker resolves


def mapped_hip_runtimes():
    """Paths of every libamdhip64 mapped into this process (Linux: /proc/self/maps)."""
    try:
        with open("/proc/self/maps") as f:
            return sorted({line.split()[-1] for line in f if "libamdhip64" in line and line.split()[-1].startswith("/")})
    except OSError:
        return []


def note_hip_use():
    """Called by the NumPy-only binding right before a call that initialises HIP (see gp_device._torch)."""
    global hip_used_before_pytorch
    import sys

    if "torch" not in sys.modules:
        hip_used_before_pytorch = True


def load():
    """Load libgpbo.so and attach prototypes.  Raises if the library or any symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpboError(
            f"{LIB_PATH} not found: build it with bayesian_optimisation_amd/csrc/build.sh "
            "(there is no CPU fallback for the acquisition path)")
    _bind_to_pytorch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    rts = mapped_hip_runtimes()
    if len({os.path.realpath(p) for p in rts}) > 1:
        # two HIP runtimes in one process do not share the device: whichever initialises second sees no GPU (seen in
        # round 1 as torch.cuda.is_available() == False after a host-pointer call).  Fail here, loudly, not later.
        raise GpboError("two different HIP runtimes are mapped into this process: " + ", ".join(rts) + ". Load PyTorch (or "
                        "this package) before any other library that links libamdhip64, so that all of them share one.")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str):
    if status != GPBO_OK:
        msg = load().gpbo_strerror(status).decode()
        raise GpboError(f"{what}: {msg} (status {status})")


def host_f64(arr):
    """ctypes pointer to a contiguous fp64 host array (kept alive by the caller)."""
    import numpy as np

    a = np.ascontiguousarray(arr, dtype=np.float64)
    return a, a.ctypes.data_as(C.c_void_p)
