"""CPU-only checks of the boundary: libgpbo.so loads without a GPU and exports every symbol that
include/gpbo.h declares; the Python binding table matches the header; the product path refuses to
run without a GPU instead of falling back to anything."""
import ctypes
import os
import re

import numpy as np
import pytest

from bayesian_optimisation_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(REPO, "include", "gpbo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpbo_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_and_loads_without_gpu():
    assert os.path.exists(_lib.LIB_PATH), "run bayesian_optimisation_amd/csrc/build.sh (or __graft_entry__.build())"
    lib = _lib.load()
    assert lib.gpbo_version() == 151
    assert lib.gpbo_padded_n(1) == 128 and lib.gpbo_padded_n(128) == 128 and lib.gpbo_padded_n(129) == 256
    assert b"workspace" in lib.gpbo_strerror(-3)


def test_every_header_symbol_is_exported_and_bound():
    names = _header_functions()
    assert len(names) >= 15
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/gpbo.h but not exported by libgpbo.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype in _lib.SIGNATURES"
    assert sorted(_lib.SIGNATURES) == names


def test_argument_validation_needs_no_gpu():
    lib = _lib.load()
    # null pointers / bad sizes are rejected on the host before anything is launched
    assert lib.gpbo_kxx_f64(None, 4, 2, None, 1e-4, 1e-6, None, 128, None) == -1
    assert lib.gpbo_posterior_workspace_bytes(100, 512, 10) == -1      # Np not a multiple of 128
    assert lib.gpbo_posterior_workspace_bytes(128, 500, 10) == -1      # chunk not a multiple of 512
    assert lib.gpbo_posterior_workspace_bytes(128, 512, 1000) > 128 * 512 * 8
    assert lib.gpbo_factorise_workspace_bytes(256) == 8 * (2 * 256 * 256 + 256 * 64 + 256)
    assert lib.gpbo_nlml_grid_max_n() == 176 and lib.gpbo_nlml_grid_wave_max_n() == 64


def test_every_compute_entry_point_rejects_null_arguments():
    """All-NULL / all-zero arguments come back as GPBO_ERR_ARG from every compute call, before any HIP call."""
    import ctypes as C

    lib = _lib.load()
    skip = {"gpbo_version", "gpbo_nlml_grid_max_n", "gpbo_nlml_grid_wave_max_n", "gpbo_gemm_f64",  # gemm: M = 0 is a valid empty product
            "gpbo_profile_create", "gpbo_profile_read", "gpbo_profile_read_kstar", "gpbo_profile_read_qei", "gpbo_profile_reset",
            "gpbo_profile_destroy"}
    checked = 0
    for name, (res, args) in _lib.SIGNATURES.items():
        if res is not C.c_int or name in skip:
            continue
        a = [None if t is C.c_void_p else (0.0 if t is C.c_double else 0) for t in args]
        assert getattr(lib, name)(*a) == -1, name
        checked += 1
    assert checked >= 15


def test_size_contracts_are_checked_on_the_host():
    """Non-null pointers but sizes that break a documented contract: rejected without touching the pointers."""
    import ctypes as C

    lib = _lib.load()
    buf = (C.c_char * 1024)()
    p = C.c_void_p((C.addressof(buf) + 255) & ~255)  # 256-byte aligned like a device allocation; never dereferenced
    ls = (C.c_double * 16)(*([0.5] * 16))
    lsp = C.cast(ls, C.c_void_p)
    # factorise: Np must be gpbo_padded_n(N); workspace must be large enough
    assert lib.gpbo_factorise_f64(p, p, 100, 2, lsp, 1e-4, 1e-6, 256, p, p, p, p, p, 1 << 40, None) == -1
    assert lib.gpbo_factorise_f64(p, p, 100, 2, lsp, 1e-4, 1e-6, 128, p, p, p, p, p, 8, None) == -3
    # append: no room left in the padding; d beyond the compiled maximum; non-positive length scale
    assert lib.gpbo_append_f64(p, p, 128, 2, lsp, 1e-4, 1e-6, 128, p, p, None, p, p, p, p, 1 << 30, None) == -1
    assert lib.gpbo_append_f64(p, p, 10, 17, lsp, 1e-4, 1e-6, 128, p, p, None, p, p, p, p, 1 << 30, None) == -1
    bad = (C.c_double * 2)(0.5, 0.0)
    assert lib.gpbo_append_f64(p, p, 10, 2, C.cast(bad, C.c_void_p), 1e-4, 1e-6, 128, p, p, None, p, p, p, p, 1 << 30,
                               None) == -1
    assert lib.gpbo_append_f64(p, p, 10, 2, lsp, 1e-4, 1e-6, 128, p, p, None, p, p, p, p, 8, None) == -3
    assert lib.gpbo_append_workspace_bytes(128) == 8 * (3 * 128 + 8)
    # posterior: chunk granule, Np granule, unknown acquisition kind
    def post(Np, chunk, kind, wbytes=1 << 40):
        return lib.gpbo_posterior_acq_f64(p, 1000, p, 100, Np, 2, lsp, p, p, 1.0, kind, 4.0, 0.0, 0.0, 0, chunk,
                                          None, None, None, p, p, wbytes, None, None)
    assert post(100, 512, 0) == -1
    assert post(128, 500, 0) == -1
    assert post(128, 512, 7) == -1
    assert post(128, 1 << 25, 0) == -1   # chunk beyond GPBO_CHUNK_MAX
    assert post(128, 512, 0, wbytes=8) == -3
    # potrf / trtri: Np a multiple of 64
    assert lib.gpbo_potrf_f64(p, 100, p, p, None) == -1
    assert lib.gpbo_trtri_f64(p, p, 100, p, p, None) == -1
    # ARD grid: N beyond the in-LDS limit
    assert lib.gpbo_nlml_grid_f64(p, p, 177, 2, p, 4, 1e-4, p, None) == -1
    assert lib.gpbo_nlml_grid_wave_f64(p, p, 65, 2, p, 4, 1e-4, p, None) == -1 and lib.gpbo_nlml_grid_wave_f64(p, p, 8, 17, p, 4, 1e-4, p, None) == -1
    # fused factorisation: S is read by 16-byte LDS-DMA pieces - an 8-byte-offset view is refused, as are an odd or short ld
    p8 = C.c_void_p(p.value + 8)
    assert lib.gpbo_cholinv_f64(p8, 512, 256, p, None, None) == -1
    assert lib.gpbo_cholinv_f64(p, 511, 256, p, None, None) == -1
    assert lib.gpbo_cholinv_f64(p, 510, 256, p, None, None) == -1
    assert lib.gpbo_cholinv_f64(p, 2 * 32896, 32896, p, None, None) == -1   # beyond the plan's cap: factorise takes the chain


def test_library_path_override(monkeypatch):
    """GPBO_LIB=/path/to/variant.so makes _lib load that file: A/B tools never overwrite the installed library."""
    import importlib
    import subprocess
    import sys

    code = (f"import sys; sys.path.insert(0, {REPO!r}); import os; os.environ['GPBO_LIB'] = '/nonexistent/variant.so'\n"
            "from bayesian_optimisation_amd import _lib\n"
            "try:\n    _lib.load(); print('loaded')\nexcept _lib.GpboError as e:\n    print('refused', '/nonexistent/variant.so' in str(e))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.stdout.strip() == "refused True", out.stdout + out.stderr
    for f in os.listdir(os.path.join(REPO, "tools")):
        if f.endswith(".sh"):
            txt = open(os.path.join(REPO, "tools", f)).read()
            assert "bayesian_optimisation_amd/libgpbo.so" not in txt.replace("cp bayesian_optimisation_amd/libgpbo.so ab_libs/", ""), \
                f"tools/{f} touches the installed library"


def test_product_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bayesian_optimisation_amd import DeviceGP, PointSelector

    with pytest.raises(_lib.GpboError):
        DeviceGP()
    ps = PointSelector()
    ps.measured_pts = np.zeros((2, 1))
    ps.measured_vals = np.zeros(2)
    ps.predicted_pts = np.zeros((5, 1))
    ps.feature_domain = [5]
    ps.length_scales = np.linspace(0.1, 1, 5)
    with pytest.raises(_lib.GpboError):
        ps.update_surrogate()


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(REPO, "bayesian_optimisation_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.replace("no CPU", ""), f"{f} mentions the oracle"


def test_python_constants_match_the_header():
    src = open(os.path.join(REPO, "include", "gpbo.h")).read()
    defs = {k: int(v) for k, v in re.findall(r"#define\s+(GPBO_[A-Z_]+)\s+\(?(-?\d+)\)?", src)}
    assert defs["GPBO_NPAD"] == _lib.NPAD and defs["GPBO_CHUNK_GRANULE"] == _lib.CHUNK_GRANULE
    assert defs["GPBO_MAX_D"] == _lib.MAX_D and defs["GPBO_ACQ_LCB"] == _lib.ACQ_LCB and defs["GPBO_ACQ_EI"] == _lib.ACQ_EI
    assert defs["GPBO_VERSION"] == _lib.load().gpbo_version()


def test_loader_refuses_a_second_hip_runtime():
    """Two HIP runtimes in one process do not share the device (round 1: torch.cuda.is_available() turned False after a
    host-pointer call through the other one).  If the system's libamdhip64 is already mapped when the package loads -
    and PyTorch's copy is what the loader binds to - loading must fail with a clear message, not at the first kernel."""
    import subprocess
    import sys

    sysrt = "/opt/rocm/lib/libamdhip64.so"
    if not os.path.exists(sysrt):
        pytest.skip("no system ROCm runtime in this image")
    code = (
        "import ctypes, sys\n"
        f"sys.path.insert(0, {REPO!r})\n"
        f"ctypes.CDLL({sysrt!r}, mode=ctypes.RTLD_GLOBAL)\n"
        "from bayesian_optimisation_amd import _lib\n"
        "try:\n"
        "    _lib.load()\n"
        "    print('loaded', len(_lib.mapped_hip_runtimes()))\n"
        "except _lib.GpboError as e:\n"
        "    print('refused' if 'two different HIP runtimes' in str(e) else 'other: ' + str(e))\n"
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1500:]
    # either PyTorch is absent / shares the system runtime (one copy mapped: fine) or the clash is reported
    assert out.stdout.strip() in ("refused", "loaded 1"), out.stdout


def test_build_from_a_clean_tree(tmp_path):
    """The build recipe of __graft_entry__.build() (csrc/build.sh) on a tree that has neither objects nor a library:
    every translation unit compiles for gfx950 and the fresh library exports every symbol of include/gpbo.h."""
    import shutil
    import subprocess

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc in this environment")
    src = os.path.join(REPO, "bayesian_optimisation_amd", "csrc")
    dst = tmp_path / "bayesian_optimisation_amd" / "csrc"
    dst.mkdir(parents=True)
    for f in os.listdir(src):
        if f.endswith((".hip", ".h", ".sh")):
            shutil.copy(os.path.join(src, f), dst / f)
    (tmp_path / "include").mkdir()
    shutil.copy(os.path.join(REPO, "include", "gpbo.h"), tmp_path / "include" / "gpbo.h")
    assert not (dst / "build").exists()
    subprocess.run(["bash", str(dst / "build.sh")], check=True, capture_output=True, timeout=900)
    lib = tmp_path / "bayesian_optimisation_amd" / "libgpbo.so"
    assert lib.exists()
    raw = ctypes.CDLL(str(lib))
    for n in _header_functions():
        assert hasattr(raw, n), n
    raw.gpbo_version.restype = ctypes.c_int
    assert raw.gpbo_version() == _lib.load().gpbo_version()
