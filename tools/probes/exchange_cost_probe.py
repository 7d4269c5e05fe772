"""What the per-sweep grid exchange of the persistent U-side solve costs: nnf_hals_solve_f32 with delta = 0 (never stops before
the budget: sweeps + publish + collect every sweep) against nnf_hals_sweeps_f32 (the same sweeps, blind) at config B's shape."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(50, 100000), (50, 131072), (30, 100000), (64, 100000)]
for r, m in cases:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.rand(300, r, device="cuda", generator=g)
    G = (A.t() @ A).contiguous()
    cross = (A.t() @ (A @ torch.rand(r, 2000, device="cuda", generator=g)))[:, torch.arange(m, device="cuda") % 2000].contiguous()
    F0 = torch.rand(r, m, device="cuda", generator=g)
    st = torch.zeros(8, dtype=torch.float64, device="cuda")
    out = []
    for name in ("solve", "blind"):
        for rep in range(3):
            F = F0.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if name == "solve":
                eng.hals_solve(cross, G, F, 100, delta=0.0, status=st)
            else:
                eng.hals_sweeps(cross, G, F, 100)
            e1.record()
            torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 10)
    print(f"r={r:3d} cols={m:7d}: persistent solve {out[0]:6.2f} us per sweep, blind sweeps {out[1]:6.2f}  (cnt {int(st[1].item())})", flush=True)
