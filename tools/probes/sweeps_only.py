"""100 blind sweeps (nnf_hals_sweeps_f32) at RxCOLS, nothing else: the subject of a rocprofv3 --pmc pass on the sweep kernels."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
for a in sys.argv[1:]:
    r, m = (int(v) for v in a.split("x"))
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.rand(300, r, device="cuda", generator=g)
    G = (A.t() @ A).contiguous()
    cross = (A.t() @ (A @ torch.rand(r, 2000, device="cuda", generator=g)))[:, torch.arange(m, device="cuda") % 2000].contiguous()
    F = torch.rand(r, m, device="cuda", generator=g)
    for _ in range(3):
        eng.hals_sweeps(cross, G, F, 100)
    torch.cuda.synchronize()
