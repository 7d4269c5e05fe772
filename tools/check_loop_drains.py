#!/usr/bin/env python3
"""Full drains of the vector-memory queue inside MFMA loops: for every kernel of a gfx950 ISA file (hipcc -S
--cuda-device-only) list the loops that hold >= 8 v_mfma and count the `s_waitcnt vmcnt(0)` between loop header and back
edge.  A streaming kernel keeps its X prefetch ring in flight across the loop; a vmcnt(0) inside it (hipcc puts one behind
loads issued under a branch whose result is used at once) stalls the wave for a whole memory round trip per iteration --
what cost the mode-2 MTTKRP 30 us of 125.  tests/test_abi_and_host.py runs `scan` over the streaming
kernels of the BASELINE configurations.
    python tools/check_loop_drains.py file.s [kernel-name substring]"""
import re
import shutil
import subprocess
import sys


def scan(path):
    """{demangled kernel name: [ {lines, mfma, vmcnt0, barriers} per loop holding >= 8 MFMAs ]}"""
    src = open(path).read().split("\n")
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
    names = [n for _, n in starts]
    dem = names
    if filt and names:
        dem = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.strip().split("\n")
    res = {}
    for (i, name), dn in zip(starts, dem):
        # (to the end of the function, not to its first s_endpgm: a kernel with an early exit has several)
        end = next((j for j in range(i, len(src)) if src[j].startswith(".Lfunc_end")), len(src))
        body = src[i:end]
        labels = {m.group(1): j for j, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        out = []
        segs = {}
        for j, l in enumerate(body):
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if m and labels.get(m.group(1), j) < j:
                a = labels[m.group(1)]
                segs[a] = min(segs.get(a, j), j)          # (several back edges to one label: the shortest trip)
        # innermost loops only: a back edge that spans another loop (an outer region the compiler laid out behind its exit
        # blocks) would count that loop's prologue waits as if they sat inside the trip
        inner = [(a, j) for a, j in segs.items() if not any(a < a2 and j2 < j for a2, j2 in segs.items())]
        for a, j in sorted(inner):
            seg = body[a:j]
            nm = sum("v_mfma" in x for x in seg)
            if nm >= 8:
                out.append(dict(lines=len(seg), mfma=nm, vmcnt0=sum("vmcnt(0)" in x for x in seg),
                                barriers=sum("s_barrier" in x for x in seg)))
        if out:
            res[dn] = out
    return res


if __name__ == "__main__":
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for dn, loops in scan(sys.argv[1]).items():
        if want in dn:
            print(dn[:110])
            for o in loops:
                print("   ", o)
