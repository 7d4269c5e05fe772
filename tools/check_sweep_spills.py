"""Build-time check: the fully unrolled HALS sweeps issue their scalar loads by hand (the destination registers are
"defined" long before the data lands), which is only safe while the register allocator never spills or copies such a
register inside a sweep.  This script compiles k_hals_fast.hip to ISA, walks every straight-line sweep block and fails
if any instruction reads or overwrites an SGPR named by a scalar load before the next s_waitcnt lgkmcnt(0) -- run it after touching
the sweep code or adding a rank instantiation.

    python tools/check_sweep_spills.py            (about two minutes)
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "nn_fac_amd", "csrc", "k_hals_fast.hip")
bad = 0
for part in range(4):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "p.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-mllvm",
                        "-pragma-unroll-threshold=4000000", f"-DHALS_PART={part}", "-I", os.path.join(ROOT, "include"), "-S",
                        "--cuda-device-only", src, "-o", out], check=True, capture_output=True)
        lines = open(out).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z15nnf_hals_kernelILi\d+ELb\dEEv9hals_args:", l)]
    for idx, (i, name) in enumerate(starts):
        end = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
        cur, blocks = [], []
        for ln in lines[i:end]:
            if re.match(r"^\.LBB", ln) or "s_cbranch" in ln or "s_branch" in ln:
                blocks.append(cur); cur = []
            else:
                cur.append(ln)
        blocks.append(cur)
        for b in blocks:
            npk = sum("v_pk_fma_f32" in x for x in b)
            if npk > 100:
                # walk the block: registers named by a scalar load are "in flight" until the next s_waitcnt lgkmcnt(0);
                # reading one of them in between (a spill, a copy, an operand) uses data that has not landed yet
                inflight, hits = set(), []
                for x in b:
                    t = x.strip()
                    m = re.match(r"s_load_dword(?:x(\d+))?\s+s\[(\d+):(\d+)\]", t) or re.match(r"s_load_dword()\s+s(\d+)()", t)
                    if m:
                        lo = int(m.group(2)); hi = int(m.group(3)) if m.group(3) else lo
                        inflight.update(range(lo, hi + 1))
                        continue
                    if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                        inflight.clear()
                        continue
                    if not inflight or t.startswith(";") or not t:
                        continue
                    ops = t.split(None, 1)[1] if " " in t or "\t" in t else ""
                    # sources AND the destination: a register reused for another value while a load into it is pending
                    # is just as fatal (the late data overwrites the new value)
                    srcs = ops.split(",")
                    used = set()
                    for o in srcs:
                        for m2 in re.finditer(r"\bs\[(\d+):(\d+)\]", o):
                            used.update(range(int(m2.group(1)), int(m2.group(2)) + 1))
                        for m2 in re.finditer(r"\bs(\d+)\b", o):
                            used.add(int(m2.group(1)))
                    if used & inflight:
                        hits.append(t)
                wl = sum("v_writelane" in x for x in b)
                flag = f"  <-- {len(hits)} use(s) of a scalar-load destination before its wait, e.g. {hits[0]}" if hits else ""
                print(f"{name}: sweep block with {npk} packed FMAs, v_writelane={wl}{flag}")
                bad += len(hits) > 0
sys.exit(1 if bad else 0)
