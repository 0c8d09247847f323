import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no libgpbo.so (it is git-ignored): build it in-tree once if hipcc is available."""
    lib = os.path.join(REPO, "bayesian_optimisation_amd", "libgpbo.so")
    if os.path.exists(lib):
        return
    import shutil
    import subprocess

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if os.path.exists(hipcc) or shutil.which("hipcc"):
        subprocess.run(["bash", os.path.join(REPO, "bayesian_optimisation_amd", "csrc", "build.sh")], check=False)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))

    return load
