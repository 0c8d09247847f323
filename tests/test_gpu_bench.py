"""bench.py as the driver runs it: `--gpus N` alone must produce an N-rank run."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *flags], capture_output=True, text=True, env=env,
                       timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "i8", "i8c", "f64b"])
def test_two_self_launched_ranks_pick_the_point_one_rank_picks_over_the_same_candidates(dtype):
    common = ["--n-obs", "384", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-also", "--dtype", dtype]
    two = _bench("--gpus", "2", "--backend", "gloo", "--all-on-device", "0", "--m-per-gpu", "65536", *common)
    one = _bench("--gpus", "1", "--m-per-gpu", "131072", *common)
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and one["n_gpus"] == 1
    assert two["ms_per_step_by_rank"]["ranks"] == 2 and two["ms_per_step_by_rank"]["max"] == two["ms_per_step"]
    assert two["ms_per_step_by_rank"]["min"] <= two["ms_per_step_by_rank"]["max"]
    assert two["config"]["candidates_total"] == one["config"]["candidates_total"] == 131072
    assert two["argmax_index"] == one["argmax_index"]
    assert two["roofline"]["frac"] > 0 and one["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_default_line_is_the_metric_configuration_and_matches_the_cpu_port():
    line = _bench("--steps", "2", "--warmup", "1", "--cpu-seconds", "4")
    assert "N=4096" in line["config"]["workload"] and "d=8" in line["config"]["workload"]
    assert line["config"]["candidates_total"] == 1 << 21 and line["dtype"] == "f64"
    assert line["cpu_baseline"]["argmax_match_on_sample"] is True
    assert line["cpu_baseline"]["sample_contains_reported_argmax"] is True      # the oracle has seen the winner itself
    assert line["cpu_baseline"]["reported_argmax_is_the_samples"] is True
    assert "N=4096" in line["cpu_baseline"]["sample"]
    # (speed ratios between the routes are reported by the line, not asserted here: a correctness suite must not turn
    #  red on a slow or shared box - tools/ab.sh compares routes on one box)
    assert 0.5 < line["roofline"]["frac"] <= 1.0
    assert line["also"]["configs[1]"]["value"] > 0 and line["also"]["ei_same_workload"]["value"] > 0
    fz = line["also"]["factorisation"]
    assert 0.0 < fz["frac"] <= 1.0 and fz["peak"] == 78.6 and abs(fz["achieved_tflops"] - fz["flop"] / fz["ms_per_call"] / 1e9) < 0.05
    c0 = line["also"]["configs[0]"]
    assert c0["index_matches_reference"] is True and c0["cpu_port_index_matches"] is True and c0["ms_per_step"] < 50
    i8 = line["also"]["int8_sliced_same_workload"]
    assert i8["argmax_matches_fp64"] is True and not i8["screen"]["fallback"] and i8["value"] > 0
    c8 = line["also"]["int8_coarse_screen_same_workload"]
    assert c8["argmax_matches_fp64"] is True and not c8["screen"]["fallback"] and c8["value"] > 0
    assert c8["screen"]["mode"] == "i8c" and 4.0 * c8["screen"]["err_max"] <= c8["screen"]["tau"]
    pb = line["also"]["prefix_bound_screen_same_workload"]
    assert pb["argmax_matches_fp64"] is True and not pb["screen"]["fallback"] and pb["value"] > 0
    assert pb["screen"]["mode"] == "bound" and pb["screen"]["rescored"] < (1 << 21) // 16
    assert pb["unit"] == "candidates disposed/s" and pb["screen"]["order"] == "fps"
    assert set(pb["survivors_by_acquisition"]) == {"lcb_explore_1", "lcb_explore_4", "lcb_explore_10", "ei"}
    assert all(v["same_point_as_plain_pass"] for v in pb["survivors_by_acquisition"].values())
    pe = line["also"]["prefix_bound_screen_ei_same_workload"]
    assert pe["argmax_matches_fp64"] is True and not pe["screen"]["fallback"] and pe["value"] > 0
    # every BASELINE config under the same clock, each with the roofline of its own dominant kernel
    c3 = line["also"]["configs[3]"]
    assert "N=8192" in c3["workload"] and c3["dtype"] == "f32" and c3["nan_count"] == 0 and c3["slice_argmax_matches_fp64"]
    assert c3["roofline"]["kernel"] == "sigma_acq_f32_kernel" and c3["roofline"]["peak"] == 157.3 and 0.3 < c3["roofline"]["frac"] <= 1.0
    assert not c3["screen"]["fallback"] and 0.0 < c3["factorisation"]["frac"] <= 1.0 and c3["kstar_roofline"]["bound"] == "hbm"
    c4 = line["also"]["configs[4]"]
    assert "qEI" in c4["workload"] and c4["nan_count"] == 0 and 0 <= c4["argmax_batch"] < (1 << 17)
    assert c4["roofline"]["kernel"] == "sigma_acq_kernel" and 0.3 < c4["roofline"]["frac"] <= 1.0
    q = c4["qei_roofline"]
    # round 5: the stage reads the Gram partials the variance launch leaves (16 x 64 doubles per batch of 8), not the rows of V
    assert q["bound"] == "hbm" and q["kernel"] == "qei_kernel" and q["bytes_per_candidate"] == 16 * 64.0 and 0.0 < q["frac"] <= 1.0
    assert q["launches"] == c4["roofline"]["launches"] == 3 * 8   # 3 timed steps x 8 chunks of 2^17
    assert c4["cpu_baseline"]["kind"] == "port" and c4["cpu_baseline"]["max_abs_diff_gpu_vs_oracle_on_sample"] <= 1e-9
    assert line["also"]["configs[1]"]["cpu_baseline"]["value"] > 0 and c3["cpu_baseline"]["value"] > 0
    assert line["roofline"]["kstar"]["kernel"] == "kstar_mu_kernel" and 0.3 < line["roofline"]["kstar"]["frac"] <= 1.0
    ag = line["also"]["ard_grid"]
    assert [e["roofline"]["cells_per_launch"] for e in ag] == [2500] * 5 and all(e["cpu_baseline"]["value"] > 0 for e in ag)
    # (N = 32, 64: the wave-per-cell kernel, vector-issue-bound; N = 176, 512, 1024: the fused kernel on the matrix cores)
    assert all(e["roofline"]["bound"] == "valu" and e["roofline"]["kernel"] == "nlml_wave_kernel" and 0.02 < e["roofline"]["valu_issue_frac"] <= 1.0 for e in ag[:2])
    assert all(e["roofline"]["bound"] == "mfma" and 0.05 < e["roofline"]["frac"] <= 1.0 for e in ag[2:])
    assert all(e["finite_cells_logdet_mode"] == 2500 and e["logdet_mode_max_rel_err_vs_oracle_on_sample"] <= 1e-9 for e in ag)
    assert all(e["reference_mode_matches_oracle_where_finite"] in (True, None) for e in ag)
    ap = line["also"]["append"]
    assert 0 < ap["append_ms"] < ap["refactorise_ms"] and ap["alpha_max_rel_diff_vs_refactorisation"] <= 1e-9
    fm = line["also"]["configs[2]_all_2^24_candidates_on_one_gpu"]
    assert fm["candidates"] == 1 << 24 and fm["equals_reduction_of_8_shard_calls"] is True and fm["nan_count"] == 0
    assert fm["shard0_is_the_headline_run"] is True and 0.8 < fm["per_candidate_rate_vs_headline"] < 1.25
    assert line["ms_per_step_by_rank"]["ranks"] == 1 and line["roofline"]["traffic_stale"] in (False, True, None)
    assert "extrapolation" in line["multi_gpu_note"]


@pytest.mark.gpu
def test_int8_sliced_mode_as_the_main_workload():
    line = _bench("--dtype", "i8", "--n-obs", "1024", "--m-per-gpu", "262144", "--steps", "2", "--warmup", "1",
                  "--cpu-seconds", "3")
    assert line["dtype"] == "i8" and line["roofline"]["kernel"] == "sigma_i8_kernel" and 0.1 < line["roofline"]["frac"] <= 1.0
    assert line["cpu_baseline"]["argmax_match_on_sample"] is True and not line["screen"]["fallback"]


@pytest.mark.gpu
def test_coarse_int8_screen_as_the_main_workload():
    line = _bench("--dtype", "i8c", "--n-obs", "1024", "--m-per-gpu", "262144", "--steps", "2", "--warmup", "1",
                  "--cpu-seconds", "3", "--no-also")
    assert line["dtype"] == "i8c" and line["roofline"]["kernel"] == "sigma_i8c_kernel" and 0.05 < line["roofline"]["frac"] <= 1.0
    assert line["roofline"]["unit"].startswith("TOP/s") and line["cpu_baseline"]["argmax_match_on_sample"] is True
    assert line["screen"]["mode"] == "i8c" and not line["screen"]["fallback"]


@pytest.mark.gpu
def test_prefix_bound_screen_as_the_main_workload():
    line = _bench("--dtype", "f64b", "--n-obs", "1024", "--m-per-gpu", "262144", "--steps", "2", "--warmup", "1",
                  "--cpu-seconds", "3", "--no-also")
    assert line["dtype"] == "f64" and "prefix-bound" in line["config"]["workload"] or "bound" in line["config"]["workload"]
    assert line["roofline"]["kernel"] == "sigma_acq_kernel" and 0.05 < line["roofline"]["frac"] <= 1.0
    assert line["cpu_baseline"]["argmax_match_on_sample"] is True
    assert line["screen"]["mode"] == "bound" and not line["screen"]["fallback"]
    kr = line["kstar_roofline"]   # the K(X*,X) interval of this route is bound by fp64 VALU issue, not by HBM
    assert kr["bound"] == "valu" and kr["peak"] == 33.0 and 0.05 < kr["frac"] <= 1.0


@pytest.mark.gpu
def test_qei_as_the_main_workload_has_its_rooflines():
    """the qEI line used to switch the kernel events off.  Now the variance launch, the K(X*,X) build and the
    qEI stage are bracketed, each on the roofline that bounds it."""
    line = _bench("--acq", "qei", "--n-obs", "512", "--m-per-gpu", "131072", "--steps", "2", "--warmup", "1", "--no-also")
    assert line["roofline"]["kernel"] == "sigma_acq_kernel" and 0.05 < line["roofline"]["frac"] <= 1.0
    assert line["qei_roofline"]["bound"] == "hbm" and 0.0 < line["qei_roofline"]["frac"] <= 1.0
    assert line["qei_roofline"]["launches"] == line["roofline"]["launches"] == 2
    assert line["kstar_roofline"]["bound"] == "hbm" and line["kstar_roofline"]["launches"] == 2
