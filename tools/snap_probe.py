"""Per-sweep cost of the V snapshots the row-sharded solve asks for (config B U side: 50 x 100000)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.engine import get_engine
eng = get_engine()
g = torch.Generator(device="cuda").manual_seed(0)
r, n, C = 50, 100000, 60
U = torch.rand(400, r, device="cuda", generator=g)
G = (U.t() @ U).contiguous()
M = torch.rand(r, n, device="cuda", generator=g) * 100
V0 = torch.rand(r, n, device="cuda", generator=g)
snap = torch.empty((C, r, n), dtype=torch.float32, device="cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, s in (("no snapshots", None), ("snapshots", snap)):
    us = t(lambda: eng.hals_sweeps(M, G, V0.clone(), C, snapshots=s))
    print(f"{name}: {us:.1f} us per {C} sweeps = {us/C:.2f} us/sweep", flush=True)
