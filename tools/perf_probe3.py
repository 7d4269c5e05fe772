"""Small-solve latency: one HALS solve of an r x n problem as the NTF / NTD drivers issue it (config D: r=30, n=500)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for r, n in ((30, 500), (50, 2000), (9, 85)):
    A = torch.rand(4 * r, r, device="cuda", generator=g)
    UtU = (A.t() @ A).contiguous()
    UtM = (A.t() @ torch.rand(4 * r, n, device="cuda", generator=g)).contiguous()
    V = torch.rand(r, n, device="cuda", generator=g)
    for k in (1, 2, 11):
        F = V.clone()
        us = timeit(lambda: eng.hals_solve(UtM, UtU, F, k, delta=0.0))
        print(f"r={r} n={n}: hals_solve {k:2d} sweeps {us:7.1f} us")
    us = timeit(lambda: eng.gram(V))
    print(f"r={r} n={n}: gram {us:7.1f} us;  torch clone {timeit(lambda: V.clone()):6.1f} us")
