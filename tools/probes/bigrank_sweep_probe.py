"""Sweeps of the generic kernel above rank 128: the column in LDS (r x 128 floats per workgroup) against the column left in
global memory (NNF_HALS_GCOL=1), microseconds per blind sweep and per sweep of a persistent solve."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
from nn_fac_amd.engine import get_engine  # noqa: E402

eng = get_engine("cuda:0")
g = torch.Generator(device="cuda").manual_seed(1)
for r, n in ((200, 2000), (200, 20000), (130, 30000), (256, 8000), (300, 2000), (400, 2000), (200, 200000)):
    A = torch.rand(3 * r, r, device="cuda", generator=g)
    G = (A.t() @ A).contiguous()
    M = (A.t() @ (A @ torch.rand(r, n, device="cuda", generator=g))).contiguous()
    V = torch.rand(r, n, device="cuda", generator=g)
    eng.hals_sweeps(M, G, V.clone(), 2)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    W = V.clone()
    a.record(); eng.hals_sweeps(M, G, W, 10); b.record(); torch.cuda.synchronize()
    blind = a.elapsed_time(b) * 100
    line = f"NNF_HALS_GCOL={os.environ.get('NNF_HALS_GCOL', '0')}  rank {r} x {n} columns: {blind:9.1f} us per blind sweep"
    if n <= 131072:
        W = V.clone()
        try:
            a.record(); st = eng.hals_solve(M, G, W, 20, delta=0.0); b.record(); torch.cuda.synchronize()
            line += f"   {a.elapsed_time(b) * 50:9.1f} us per sweep of a 20-sweep solve"
        except Exception as e:   # noqa: BLE001
            line += f"   solve: {type(e).__name__}"
    print(line, flush=True)
