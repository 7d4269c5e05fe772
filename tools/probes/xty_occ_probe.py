"""W^T X at ranks 96 ... 128 compiled for one resident workgroup per CU (the product: up to 348 registers) against two
(-DXTY_BIG_WG=2, tools/abl_build.sh: <= 256 registers): launch time, result against float64 on a column sample."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
from nn_fac_amd.engine import get_engine  # noqa: E402

eng = get_engine("cuda:0")
g = torch.Generator(device="cuda").manual_seed(2)
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(1000000, 4000, 100), (125000, 4000, 100), (1000000, 2000, 96), (1000000, 2000, 128)]
for m, n, r in shapes:
    X = torch.rand(m, n, device="cuda", generator=g)
    Ut = torch.rand(r, m, device="cuda", generator=g)
    out = torch.empty(r, n, device="cuda")
    for _ in range(3):
        eng.xty(X, Ut, out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); eng.xty(X, Ut, out=out); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    want = Ut.double() @ X[:, :64].double()
    err = float((out[:, :64].double() - want).norm() / want.norm())
    print(f"{os.environ.get('NNF_LIBRARY', 'product')[-24:]:24s} {m}x{n} rank {r}: median {t[5]:.3f} ms (min {t[0]:.3f}; slab reduction included)  "
          f"rel err {err:.1e}", flush=True)
    del X, Ut, out
