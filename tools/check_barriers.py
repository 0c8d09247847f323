"""Static check of the compiled kernels: no workgroup barrier may be reached with LDS operations of the same wave still in
flight.  `__syncthreads()` is a fence + `s_barrier`, and the compiler is expected to put `s_waitcnt lgkmcnt(0)` in front of every
`s_barrier` that LDS reads or writes can reach - round 5 found hipcc leaving it out at a LOOP HEADER (ard.hip's in-LDS
likelihood kernel: the ds_write of the previous trip still in flight when the other waves passed the barrier; 2 % of the cells
wrong at d = 16 on a full chip, found by tools/fuzz_ard.py), which is why the kernels call gpbo_syncthreads() (an explicit
`s_waitcnt vmcnt(0) lgkmcnt(0)` + the barrier) wherever a barrier sits in a loop.  This tool looks for the pattern itself:
it compiles every translation unit to gfx950 assembly, builds each kernel's control-flow graph and propagates "an LDS WRITE
(ds_write*, LDS atomics) may be outstanding" forwards; `s_waitcnt` with lgkmcnt(0) clears it; an `s_barrier` reached with it
set is reported.  Not checked, because the pipelined kernels do it on purpose: LDS READS in flight across a barrier (operand
fragments of the next step fetched early from a ring stage that nobody refills until a later barrier - sigma_acq.hip,
posterior_f32.hip, ozaki.hip, cholinv.hip: 35 such barriers) and LDS-DMA loads (global_load_lds_*: counted by vmcnt, waited
for stage by stage).
usage: python tools/check_barriers.py [unit ...]      exit code 1 if anything is reported"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bayesian_optimisation_amd", "csrc")
UNITS = ["kernel_build", "kstar_mfma", "gemm_f64", "factor", "cholinv", "subset", "update", "sigma_acq", "ard",
         "ard_wave", "posterior_f32", "rescore", "ozaki"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def assemble(unit, outdir):
    out = os.path.join(outdir, unit + ".s")
    extra = ["-mllvm", "-amdgpu-kernarg-preload-count=16"] if unit == "cholinv" else []   # as build.sh does
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *extra, "-S",
                    "--cuda-device-only", os.path.join(CSRC, unit + ".hip"), "-o", out], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$")
LDS_WRITE = re.compile(r"^ds_(write|add|sub|rsub|inc|dec|min|max|and|or|xor|mskor|cmpst|wrxchg|wrap|pk_add|append|consume)")
BR = re.compile(r"^(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)")


def clears_lgkm(ins):
    # "s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(0) lgkmcnt(0)"; a partial wait (lgkmcnt(n > 0)) clears nothing
    return ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins


def check_function(name, lines):
    blocks, cur, order = {"entry": []}, "entry", ["entry"]
    for ln in lines:
        m = LABEL.match(ln)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        blocks[cur].append(s.split(";")[0].strip())
    succ = {b: [] for b in order}
    for i, b in enumerate(order):
        fall = True
        for ins in blocks[b]:
            m = BR.match(ins)
            if m:
                succ[b].append(m.group(2))
                if m.group(1) == "s_branch":
                    fall = False
            if ins.startswith("s_endpgm") or ins.startswith("s_setpc"):
                fall = False
        if fall and i + 1 < len(order):
            succ[b].append(order[i + 1])

    def run(b, st, report=None):
        for k, ins in enumerate(blocks[b]):
            if LDS_WRITE.match(ins):
                st = True
            elif clears_lgkm(ins):
                st = False
            elif ins.startswith("s_barrier") and st and report is not None:
                report.append((name, b, k))
        return st

    inn = {b: False for b in order}
    changed = True
    while changed:   # forward "may be outstanding" to a fixed point
        changed = False
        for b in order:
            if run(b, inn[b]):
                for t in succ[b]:
                    if t in inn and not inn[t]:
                        inn[t] = True
                        changed = True
    found = []
    for b in order:
        run(b, inn[b], found)
    return found


def check_file(path):
    found, name, lines, nbar = [], None, [], 0
    for ln in open(path):
        m = FUNC.match(ln)
        if m and not LABEL.match(ln):
            if name:
                found += check_function(name, lines)
            name, lines = m.group(1), []
            continue
        if name:
            lines.append(ln)
            if re.match(r"^\s+s_barrier", ln):
                nbar += 1
    if name:
        found += check_function(name, lines)
    return found, nbar


def main(units):
    with tempfile.TemporaryDirectory() as td:
        with ThreadPoolExecutor(max_workers=min(8, len(units))) as ex:
            paths = list(ex.map(lambda u: assemble(u, td), units))
        bad, total = [], 0
        for u, p in zip(units, paths):
            f, n = check_file(p)
            total += n
            bad += [(u,) + x for x in f]
    for u, fn, blk, k in bad:
        print(f"{u}.hip: {fn}: s_barrier in block {blk} (instruction {k}) reachable with an LDS operation outstanding")
    print(f"check_barriers: {total} barriers in {len(units)} translation units, {len(bad)} reachable with LDS operations in flight")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or UNITS))
