"""Build-time ISA check of the hand-scheduled HALS sweep kernels (run by tests/test_abi_and_host.py and usable alone).

The fully unrolled sweeps issue loads by hand: the destination registers are "defined" for the compiler long before the data
lands, which is only safe while the register allocator never reads, copies, spills or reuses such a register between the
load and the wait that covers it.  hipcc cannot see these loads, so nothing but the emitted ISA can tell.

  k_hals_fast.hip (one lane per column): s_load_dword* into SGPRs; scalar loads return out of order, the only wait is
      s_waitcnt lgkmcnt(0).  Every straight-line sweep block (> 100 v_pk_fma_f32) is walked: any instruction that names
      an SGPR with a load in flight -- as source or destination -- before the next lgkmcnt(0) is a violation.
  k_hals_quad.hip (four lanes per column): ds_read_b128 into VGPRs two rows ahead; LDS returns in order, so
      s_waitcnt lgkmcnt(N) retires all but the N youngest reads.  Same walk over the blocks with > 16 ds_read_b128.

    python tools/check_sweep_spills.py            (seconds after `make`: the build keeps each unit's ISA listing; without
                                                   them the eight translation units are compiled to assembly, ~3 minutes)

Exit code 1 / a non-empty list from check_all() = a rank instantiation (or a compiler bump) broke the invariant: the
abort class of round 1 (gpurun_out/t44.log: a row-split sweep over the SGPR budget).
"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nn_fac_amd", "csrc")


def _isa(src, define):
    """ISA of one translation unit: the listing the build left next to its object (same compilation as the shipped code,
    nn_fac_amd/csrc/Makefile) when it is newer than every source, else a fresh compilation to assembly."""
    part = define.split("=")[1]
    kept = os.path.join(CSRC, "build", f"{os.path.splitext(src)[0]}{part}.s")
    deps = [os.path.join(CSRC, f) for f in (src, "k_hals_common.h", "nnf_internal.h")] + \
        [os.path.join(ROOT, "include", "nnfac_hip.h")]
    if os.path.exists(kept) and os.path.getmtime(kept) >= max(os.path.getmtime(d) for d in deps):
        return open(kept).read().split("\n")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "p.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-mllvm",
                        "-pragma-unroll-threshold=4000000", define, "-I", os.path.join(ROOT, "include"), "-S",
                        "--cuda-device-only", os.path.join(CSRC, src), "-o", out], check=True, capture_output=True)
        return open(out).read().split("\n")


def _kernels(lines, pattern):
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(pattern, l)]
    for idx, (i, name) in enumerate(starts):
        end = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
        cur, blocks = [], []
        for ln in lines[i:end]:
            if re.match(r"^\.LBB", ln) or "s_cbranch" in ln or "s_branch" in ln:
                blocks.append(cur)
                cur = []
            else:
                cur.append(ln)
        blocks.append(cur)
        yield name, blocks


def _regs(text, bank):
    used = set()
    for m in re.finditer(r"\b%s\[(\d+):(\d+)\]" % bank, text):
        used.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\b%s(\d+)\b" % bank, text):
        used.add(int(m.group(1)))
    return used


def check_fast(part):
    """Violations in k_hals_fast.hip, translation unit `part`: list of (kernel, instruction)."""
    lines = _isa("k_hals_fast.hip", f"-DHALS_PART={part}")
    bad, seen = [], 0
    for name, blocks in _kernels(lines, r"^_Z15nnf_hals_kernelILi\d+ELb\dEEv9hals_args:"):
        for b in blocks:
            if sum("v_pk_fma_f32" in x for x in b) <= 100:
                continue
            seen += 1
            inflight = set()
            for x in b:
                t = x.strip()
                m = re.match(r"s_load_dword(?:x\d+)?\s+(s\[\d+:\d+\]|s\d+)", t)
                if m:
                    inflight |= _regs(m.group(1), "s")
                    continue
                if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                    inflight.clear()
                    continue
                if not inflight or t.startswith(";") or not t or " " not in t.replace("\t", " "):
                    continue
                if _regs(t.split(None, 1)[1], "s") & inflight:
                    bad.append((name, t))
    return bad, seen


def check_quad(part):
    """Violations in k_hals_quad.hip, translation unit `part`."""
    lines = _isa("k_hals_quad.hip", f"-DQUAD_PART={part}")
    bad, seen = [], 0
    for name, blocks in _kernels(lines, r"^_Z20nnf_hals_quad_kernelILi\d+EEv9hals_args:"):
        for b in blocks:
            if sum("ds_read_b128" in x for x in b) <= 16:
                continue
            seen += 1
            queue = []          # in-order list of destination register sets of the LDS reads in flight
            for x in b:
                t = x.strip()
                m = re.match(r"ds_read_b\d+\s+(v\[\d+:\d+\]|v\d+)", t)
                if m:
                    queue.append(_regs(m.group(1), "v"))
                    continue
                m = re.match(r"s_waitcnt.*lgkmcnt\((\d+)\)", t)
                if m:
                    keep = int(m.group(1))
                    queue = queue[len(queue) - keep:] if keep else []
                    continue
                if t.startswith("s_waitcnt") and "lgkmcnt" not in t:
                    continue
                if not queue or t.startswith(";") or not t or " " not in t.replace("\t", " "):
                    continue
                fl = set().union(*queue)
                if _regs(t.split(None, 1)[1], "v") & fl:
                    bad.append((name, t))
    return bad, seen


def check_all(verbose=False):
    jobs = [(check_fast, p) for p in range(4)] + [(check_quad, p) for p in range(4)]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        res = list(ex.map(lambda j: j[0](j[1]), jobs))
    bad, blocks = [], 0
    for (fn, p), (b, seen) in zip(jobs, res):
        blocks += seen
        bad += b
        if verbose:
            print(f"{fn.__name__}({p}): {seen} sweep blocks, {len(b)} violation(s)" + (f", e.g. {b[0]}" if b else ""))
    return bad, blocks


if __name__ == "__main__":
    bad, blocks = check_all(verbose=True)
    print(f"{blocks} sweep blocks checked, {len(bad)} violation(s)")
    sys.exit(1 if bad or blocks == 0 else 0)
