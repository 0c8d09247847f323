"""Which kernel sources a committed PMC pass (profiles/pmc_sigma_acq.json) was collected on.

bench.py replays `roofline.traffic` and `kstar_roofline.valu` from the committed rocprofv3 --pmc passes (counters cannot be
read from inside an un-profiled run).  A kernel change that alters traffic would go unnoticed until profiles/collect.sh is
run again - so summarise.py stores the hash of the sources it profiled in the entry, bench.py reports
`traffic_stale: true` when today's sources differ, and tests/test_host_logic_cpu.py fails until the pass is re-collected."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesian_optimisation_amd", "csrc")
COMMON = ["kernel_build.hip", "exp_neg.h", "gpbo_internal.h"]
SOURCES = {"f64": ["sigma_acq.hip"] + COMMON, "f64b": ["sigma_acq.hip", "kstar_mfma.hip", "rescore.hip"] + COMMON,
           "f32": ["posterior_f32.hip"] + COMMON, "i8": ["ozaki.hip"] + COMMON, "i8c": ["ozaki.hip"] + COMMON,
           "ard": ["ard.hip", "potrf_diag64.h", "exp_neg.h", "gpbo_internal.h"],
           "ard_wave": ["ard_wave.hip", "potrf_diag64.h", "exp_neg.h", "gpbo_internal.h"]}


def kernel_source_hash(dtype: str) -> str:
    h = hashlib.sha256()
    for name in SOURCES.get(dtype, SOURCES["f64"]):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]
