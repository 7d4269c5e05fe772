"""Per-kernel timing at a given shape (device-generated data).  Usage: python tools/perf_probe.py [m n r]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.engine import get_engine  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps  # ms


def main():
    m, n, r = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (100000, 2000, 50)
    eng = get_engine("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(0)
    W, H = torch.rand(m, r, device="cuda", generator=g), torch.rand(r, n, device="cuda", generator=g)
    X = W @ H + 1e-2 * torch.rand(m, n, device="cuda", generator=g)
    Ut = torch.rand(r, m, device="cuda", generator=g)
    V = torch.rand(r, n, device="cuda", generator=g)
    xbytes = m * n * 4
    flops = 2.0 * r * m * n
    print(f"shape m={m} n={n} r={r}  X={xbytes/1e6:.0f} MB")
    for name, fn, fl, by in (
        ("xty (UtX)", lambda: eng.xty(X, Ut), flops, xbytes + r * m * 4),
        ("xht (VXt)", lambda: eng.xht(X, V), flops, xbytes + r * m * 4),
        ("frob", lambda: eng.frob_resid(X, Ut, V), flops, xbytes + r * m * 4),
        ("gram Ut", lambda: eng.gram(Ut), 2.0 * r * r * m, r * m * 4),
        ("gram V", lambda: eng.gram(V), 2.0 * r * r * n, r * n * 4),
    ):
        ms = timeit(fn)
        print(f"{name:12s} {ms*1e3:9.1f} us   {fl/ms/1e9:8.2f} TF/s   {by/ms/1e6:8.1f} GB/s")
    # HALS sweeps (fixed count): U side (r x m) and V side (r x n)
    UtM, UtU = eng.xht(X, V), eng.gram(V)
    for k in (1, 10):
        F = Ut.clone()
        ms = timeit(lambda: eng.hals_sweeps(UtM, UtU, F, k), reps=10)
        print(f"hals_sweeps U-side x{k:<3d} {ms*1e3:9.1f} us  ({ms*1e3/k:7.1f} us/sweep)")
    F = Ut.clone()
    ms = timeit(lambda: eng.hals_solve(UtM, UtU, F, 10, delta=0.0), reps=10)
    print(f"hals_solve U-side 10 sweeps (grid barrier) {ms*1e3:9.1f} us ({ms*1e2:7.1f} us/sweep)")
    F = Ut.clone()
    ms = timeit(lambda: eng.hals_sweeps(UtM, UtU, F, 100), reps=5)
    print(f"hals_sweeps U-side x100      {ms*1e3:9.1f} us  ({ms*10:7.1f} us/sweep)")
    F = Ut.clone()
    ms = timeit(lambda: eng.hals_solve(UtM, UtU, F, 100, delta=0.0), reps=5)
    print(f"hals_solve U-side x100       {ms*1e3:9.1f} us  ({ms*10:7.1f} us/sweep)")
    VtM, VtV = eng.xty(X, Ut), eng.gram(Ut)
    for k in (10, 100):
        F = V.clone()
        ms = timeit(lambda: eng.hals_solve(VtM, VtV, F, k, delta=0.0), reps=5)
        print(f"hals_solve V-side x{k:<3d} {ms*1e3:9.1f} us  ({ms*1e3/k:7.1f} us/sweep)")
    F = V.clone()
    ms = timeit(lambda: eng.hals_sweeps(VtM, VtV, F, 100), reps=5)
    print(f"hals_sweeps V-side x100 (no exchange) {ms*1e3:9.1f} us  ({ms*10:7.1f} us/sweep)")
    import os
    os.environ["NNF_HALS_FORCE"] = "lane"
    F = V.clone()
    ms = timeit(lambda: eng.hals_solve(VtM, VtV, F, 100, delta=0.0), reps=5)
    print(f"hals_solve V-side x100 lane layout {ms*1e3:9.1f} us  ({ms*10:7.1f} us/sweep)")
    del os.environ["NNF_HALS_FORCE"]
    # full NMF iterations
    from nn_fac_amd.nmf import compute_nmf
    U0 = torch.rand(m, r, device="cuda", generator=g)
    V0 = torch.rand(r, n, device="cuda", generator=g)
    sw = []
    compute_nmf(X, r, U0, V0, n_iter_max=2, tol=0, deterministic=True)
    torch.cuda.synchronize()
    t0 = time.time()
    U, Vv, costs, toc = compute_nmf(X, r, U0, V0, n_iter_max=10, tol=0, return_costs=True, deterministic=True,
                                    sweep_log=sw)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"NMF HALS 10 iterations: {dt*100:.2f} ms/iter  -> {10/dt:.1f} it/s   sweeps={sw}")
    print("costs", [f"{c:.4e}" for c in costs])


if __name__ == "__main__":
    main()
