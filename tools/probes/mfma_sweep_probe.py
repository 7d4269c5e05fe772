"""Lane-per-column kernel against the MFMA push-form kernel (k_hals_mfma.hip): blind sweeps (nnf_hals_sweeps_f32), results against an
fp64 Gauss-Seidel reference on a slice of the columns, microseconds per sweep.  Usage: mfma_sweep_probe.py [RxCOLS ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from nn_fac_amd.engine import get_engine

eng = get_engine("cuda:0")
cases = [(int(a), int(b)) for a, b in (x.split("x") for x in sys.argv[1:])] or [(100, 125000), (96, 125000), (80, 125000), (64, 100000), (50, 100000), (48, 100000)]


def ref_sweeps(G, B, V, ns):
    G, B, V = G.double().cpu().numpy(), B.double().cpu().numpy(), V.double().cpu().numpy().copy()
    nds = []
    for _ in range(ns):
        nd = 0.0
        for k in range(G.shape[0]):
            if G[k, k] != 0:
                d = np.maximum((B[k] - G[k] @ V) / G[k, k], -V[k])
                V[k] += d
                nd += float((d * d).sum())
        nds.append(nd)
    return V, nds


for r, m in cases:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.rand(300, r, device="cuda", generator=g)
    G = (A.t() @ A).contiguous()
    cross = (A.t() @ (A @ torch.rand(r, 2000, device="cuda", generator=g)))[:, torch.arange(m, device="cuda") % 2000].contiguous()
    cross = cross * (1 + 0.01 * torch.rand(r, m, device="cuda", generator=g))
    F0 = torch.rand(r, m, device="cuda", generator=g)
    out = {}
    for kind in ("lane", "mfma"):
        os.environ["NNF_HALS_FORCE"] = kind
        for ns in (1, 6):
            F = F0.clone()
            nd = eng.hals_sweeps(cross, G, F, ns)
            torch.cuda.synchronize()
            out[(kind, ns)] = (F, nd.clone())
        ns = 20
        F = F0.clone()
        eng.hals_sweeps(cross, G, F, 3)
        torch.cuda.synchronize()
        F = F0.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nd = eng.hals_sweeps(cross, G, F, ns)
        e1.record()
        torch.cuda.synchronize()
        out[(kind, "us")] = e0.elapsed_time(e1) * 1e3 / ns
    sl = slice(0, 512)
    Vr, ndr = ref_sweeps(G, cross[:, sl], F0[:, sl], 6)
    line = f"r={r:4d} cols={m:7d}: lane {out[('lane', 'us')]:7.2f} us/sweep  mfma {out[('mfma', 'us')]:7.2f} us/sweep |"
    for ns in (1, 6):
        Fl, ndl = out[("lane", ns)]
        Fm, ndm = out[("mfma", ns)]
        rel = float((Fl - Fm).norm() / Fl.norm())
        line += f" ns={ns}: rel(lane,mfma)={rel:.2e} nd {float(ndl[-1]):.6e}/{float(ndm[-1]):.6e}"
    Fl, Fm = out[("lane", 6)][0][:, sl].double().cpu().numpy(), out[("mfma", 6)][0][:, sl].double().cpu().numpy()
    line += f" | vs fp64 (6 sweeps, 512 cols): lane {np.linalg.norm(Fl - Vr) / np.linalg.norm(Vr):.2e} mfma {np.linalg.norm(Fm - Vr) / np.linalg.norm(Vr):.2e}"
    print(line, flush=True)
