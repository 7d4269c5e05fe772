import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from nn_fac_amd.engine import get_engine
from nn_fac_amd import nmf as nmf_mod
dev = torch.device("cuda:0"); eng = get_engine(dev)
m, n, r = 1000000, 4000, 100
parts = [bench.synth_nmf_block_device(m // 8, n, r, b, 977, dev, torch) for b in range(8)]
X = torch.cat([p[0] for p in parts]); Ut = torch.cat([p[1] for p in parts]).t().contiguous(); del parts
V = torch.rand(r, n, device=dev, generator=torch.Generator(device=dev).manual_seed(4242))
print("cross_rounding", eng.cross_rounding(X, Ut))
ws = nmf_mod._StepBuffers(X, r)
costs = []
def retired(it, cost, sw):
    costs.append((it, float(cost), sw, ws.direct_cost, [float(x) for x in ws.host[ws.slot][19:22]] if hasattr(ws, "host") else None))
    return False
retired.revise_last = lambda c: print("revise_last", c)
nmf_mod.run_steps(eng, ws, X, r, Ut, V, 8, "hals", 2, [None, None], [], [False, False], True, retired)
for c in costs: print(c)
print("cal", ws.cross_rounding, "direct", ws.direct_cost)
