"""No workgroup barrier of the shipped kernels is reachable with an LDS write of the same wave still in flight
(tools/check_barriers.py: the compiled gfx950 assembly of every translation unit, control-flow graph per kernel).  Round 5
found hipcc omitting `s_waitcnt lgkmcnt(0)` in front of a barrier at a loop header (ard.hip, in-LDS likelihood kernel)."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import check_barriers as cb  # noqa: E402

LOOP_HEADER_BARRIER = """
kernel_a:
	s_waitcnt lgkmcnt(0)
	s_barrier
.LBB0_1:
	s_barrier
	ds_read_b64 v[0:1], v2
	s_waitcnt lgkmcnt(0)
	v_add_f64 v[0:1], v[0:1], v[0:1]
	ds_write_b64 v2, v[0:1]
	s_cbranch_scc1 .LBB0_1
	s_endpgm
.Lfunc_end0:
kernel_b:
.LBB1_1:
	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_barrier
	ds_read_b64 v[0:1], v2
	s_waitcnt lgkmcnt(0)
	ds_write_b64 v2, v[0:1]
	s_cbranch_scc1 .LBB1_1
	s_endpgm
.Lfunc_end1:
kernel_c:
.LBB2_1:
	ds_read_b64 v[0:1], v2
	s_waitcnt vmcnt(0)
	s_barrier
	s_waitcnt lgkmcnt(0)
	s_cbranch_scc1 .LBB2_1
	s_endpgm
.Lfunc_end2:
"""


def test_checker_sees_a_write_in_flight_at_a_loop_header_barrier(tmp_path):
    p = tmp_path / "k.s"
    p.write_text(LOOP_HEADER_BARRIER)
    found, nbar = cb.check_file(str(p))
    assert nbar == 4
    # kernel_a: the write of the previous trip reaches the loop header's barrier; kernel_b waits first; kernel_c leaves only
    # a READ in flight (the pipelined kernels' pattern: not reported)
    assert [(f[0], f[1]) for f in found] == [("kernel_a", ".LBB0_1")]


@pytest.mark.skipif(shutil.which(cb.HIPCC) is None and not os.path.exists(cb.HIPCC), reason="hipcc not installed")
def test_no_barrier_of_the_shipped_kernels_is_reachable_with_an_lds_write_in_flight(capsys):
    rc = cb.main(cb.UNITS)
    out = capsys.readouterr().out
    assert rc == 0, out
    assert "0 reachable" in out


@pytest.mark.skipif(shutil.which(cb.HIPCC) is None and not os.path.exists(cb.HIPCC), reason="hipcc not installed")
def test_the_wave_per_cell_likelihood_kernel_keeps_its_matrix_in_registers(tmp_path):
    """csrc/ard_wave.hip holds a cell's matrix in registers; its register allocation proved fragile (one conditional load
    written differently spilled 124 registers at NMAX = 32 and the launch moved 157 MB of scratch, 0.026 -> 0.045 ms): no
    instance may need more than a few bytes of scratch."""
    import re

    s = open(cb.assemble("ard_wave", str(tmp_path))).read()
    sizes = {re.search(r"\.name:\s+(\S+)", b).group(1): int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1))
             for b in s.split("  - .agpr_count:")[1:]}
    waves = {k: v for k, v in sizes.items() if "nlml_wave_kernel" in k}
    assert len(waves) == 16, sorted(sizes)
    assert max(waves.values()) <= 16, waves
