"""X H^T at rank 100 with four row tiles per wave (404 registers, one wave per SIMD) against two (NNF_XHT_NT2=1: 248 registers, two
workgroups per CU): launch time at 10^6 x 4000 and 125000 x 4000, result against a float64 product of row samples."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
from nn_fac_amd.engine import get_engine  # noqa: E402

eng = get_engine("cuda:0")
g = torch.Generator(device="cuda").manual_seed(2)
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(1000000, 4000, 100), (125000, 4000, 100), (300000, 2000, 80)]
for m, n, r in shapes:
    X = torch.rand(m, n, device="cuda", generator=g)
    V = torch.rand(r, n, device="cuda", generator=g)
    out = torch.empty(r, m, device="cuda")
    for _ in range(3):
        eng.xht(X, V, out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); eng.xht(X, V, out=out); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    lo = m - 1000
    want = V.double() @ X[lo:].double().t()
    err = float((out[:, lo:].double() - want).norm() / want.norm())
    print(f"NNF_XHT_NT2={os.environ.get('NNF_XHT_NT2', '0')}  {m}x{n} rank {r}: median {t[5]:.3f} ms (min {t[0]:.3f})  "
          f"{2.0 * m * n * r / t[5] / 1e9 / 157.3:.3f} of the fp32 MFMA peak   rel err {err:.1e}", flush=True)
    del X, V, out
