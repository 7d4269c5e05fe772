"""Does `rocprofv3 --kernel-trace` change how long a kernel RUNS, or only how it is measured?  The same stand-alone launches
(config C's right / left MU updates and config B's W^T X at 100000 x 2000 rank 50) timed with HIP events on the launch stream,
once in a plain process and once under the profiler (whose own per-launch durations are in its kernel trace):

    python tools/probes/profiler_effect_probe.py                                   # plain
    rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/probes/profiler_effect_probe.py   # same launches, profiled
Also prints the shader clock the driver reports while the launches run (rocm-smi, best effort)."""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nn_fac_amd.engine import get_engine  # noqa: E402

eng = get_engine("cuda:0")
g = torch.Generator(device="cuda").manual_seed(1)
m, n, r = 100000, 2000, 50
X = torch.rand(m, n, device="cuda", generator=g) + 0.01
Ut = torch.rand(r, m, device="cuda", generator=g)
V = torch.rand(r, n, device="cuda", generator=g)
out_v, out_u, out_x = torch.empty_like(V), torch.empty_like(Ut), torch.empty(r, n, device="cuda")


def timed(fn, reps=30, gap_s=0.0):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
        if gap_s:
            torch.cuda.synchronize()
            time.sleep(gap_s)
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[0], t[-1]


def sclk():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        return [ln.strip() for ln in o.splitlines() if "sclk" in ln.lower()][:1]
    except Exception as e:   # noqa: BLE001
        return [repr(e)]


spin = torch.empty(64 << 20, device="cuda")
for _ in range(300):
    spin.mul_(1.0001)
torch.cuda.synchronize()
for name, fn in (("mu_right beta=1", lambda: eng.mu_right(X, Ut, V, 1, out=out_v)), ("mu_left beta=1", lambda: eng.mu_left(X, Ut, V, 1, out=out_u)),
                 ("xty", lambda: eng.xty(X, Ut, out=out_x))):
    med, lo, hi = timed(fn)
    med2, lo2, hi2 = timed(fn, reps=10, gap_s=0.05)
    print(f"{name:18s} back to back: median {med:7.1f} us (min {lo:7.1f}, max {hi:7.1f});  with 50 ms idle gaps: median {med2:7.1f} (min {lo2:7.1f}, max {hi2:7.1f})",
          flush=True)
print("clock while busy:", sclk(), flush=True)
