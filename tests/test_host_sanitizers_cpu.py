"""the host side of libgpbo under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU.

tools/sanitize_host.sh compiles every translation unit of csrc/ with the host code instrumented and links
tools/sanitize_host.cpp against it: the launch planner of the fused factorisation (cholinv_plan.h) for every padded size
128 ... 16,384 and the option sets the tests use, and the argument-validation table of tests/test_abi_cpu.py - everything
that runs before a kernel is launched.  No GPU is needed or used (sanitizers stay on the CPU build)."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_planner_and_argument_validation_under_asan_ubsan(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc in this environment")
    out = subprocess.run(["bash", os.path.join(REPO, "tools", "sanitize_host.sh"), str(tmp_path)], capture_output=True,
                         text=True, timeout=1500)
    tail = (out.stdout + out.stderr)[-4000:]
    assert out.returncode == 0, tail
    assert "sanitize_host ok:" in out.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
