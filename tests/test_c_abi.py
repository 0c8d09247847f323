"""include/gpbo.h from C: the header compiles as strict C99 (CPU), and a plain-C program linked against libgpbo.so
reproduces the Python path's numbers (GPU)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "c", "abi_smoke.c")
INC = os.path.join(REPO, "include")
LIBDIR = os.path.join(REPO, "bayesian_optimisation_amd")


def _gcc():
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    return gcc


def test_header_is_strict_c99(tmp_path):
    out = subprocess.run([_gcc(), "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", INC, "-c", SRC, "-o",
                          str(tmp_path / "abi_smoke.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


@pytest.mark.gpu
def test_plain_c_caller_matches_the_python_path(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    out = subprocess.run([_gcc(), "-std=c99", "-I", INC, SRC, "-o", exe, "-L", LIBDIR, "-lgpbo",
                          f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=240)
    assert run.returncode == 0, run.stderr
    lines = run.stdout.strip().splitlines()
    info, idx, nan, best = lines[0].split()
    vals = np.array([[float(v) for v in ln.split()] for ln in lines[1:]])
    from bayesian_optimisation_amd import DeviceGP

    X = np.array([0.0, 1.0, 2.5, 4.0]).reshape(-1, 1)
    y = np.array([1.0, -0.5, 0.25, 2.0])
    Xs = (0.5 * np.arange(9)).reshape(-1, 1)
    r = DeviceGP(chunk=512).factorise(X, y, [0.8]).score(Xs, dense=True)
    assert int(info) == 0 and int(nan) == 0 and int(idx) == r.best_idx and float(best) == r.best_val
    assert np.array_equal(vals[:, 0], r.mu.cpu().numpy()) and np.array_equal(vals[:, 1], r.sigma.cpu().numpy())
    assert np.array_equal(vals[:, 2], r.acq.cpu().numpy())
