"""Rounding of an MTTKRP entry at config D's shape (500^3, rank 30) against float64: relative rms and MEAN per mode -- what the
Gram-identity cost of the NTF loop (ntf.py:462-470 form) assumes as 6e-8 / 0 (nnf_nmf_gram_cost_f32's defaults)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
from nn_fac_amd.engine import get_engine  # noqa: E402

eng = get_engine("cuda:0")
g = torch.Generator(device="cuda").manual_seed(3)
for (I, J, K, R) in ((500, 500, 500, 30), (300, 200, 1000, 64)):
    F = [torch.rand(d, R, device="cuda", generator=g) for d in (I, J, K)]
    T = (torch.einsum("ir,jr,kr->ijk", *F) * (1 + 0.05 * torch.rand(I, J, K, device="cuda", generator=g))).contiguous()
    Ft = [f.t().contiguous() for f in F]
    T64 = T.double()
    for mode in (0, 1, 2):
        got = eng.mttkrp3(T, Ft, mode).double()
        a, b = [F[m].double() for m in range(3) if m != mode]
        sub = {0: "ijk,jr,kr->ri", 1: "ijk,ir,kr->rj", 2: "ijk,ir,jr->rk"}[mode]
        want = torch.einsum(sub, T64, a, b)
        rel = (got - want) / want
        print(f"{I}x{J}x{K} rank {R} mode {mode}: relative rms {float(rel.pow(2).mean().sqrt()):.2e}  mean {float(rel.mean()):+.2e}  max {float(rel.abs().max()):.2e}",
              flush=True)
    Y = eng.ttm3(T, Ft[2], 2)
    for axis, other, sub in ((2, 1, "ijk,jr,kr->ri"), (1, 0, "ijk,ir,kr->rj")):
        got = eng.mttkrp3_from_partial(Y, Ft[other], axis).double()
        a, b = (F[1].double(), F[2].double()) if axis == 2 else (F[0].double(), F[2].double())
        want = torch.einsum(sub, T64, a, b)
        rel = (got - want) / want
        print(f"   through the partial product, axis {axis}: relative rms {float(rel.pow(2).mean().sqrt()):.2e}  mean {float(rel.mean()):+.2e}", flush=True)
    del T, T64
