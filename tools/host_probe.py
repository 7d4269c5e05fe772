"""Host-side cost of enqueuing one outer iteration (no device wait) vs its device time: NMF config B and NTF config D."""
import math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.engine import get_engine
from nn_fac_amd import nmf as nm, ntf as nt
eng = get_engine()
g = torch.Generator(device="cuda").manual_seed(0)
# NTF D
I = 500; R = 30
F = [torch.rand(I, R, device="cuda", generator=g) for _ in range(3)]
T = torch.einsum('ir,jr,kr->ijk', *F) + 1e-2 * torch.rand(I, I, I, device="cuda", generator=g)
Ft = [torch.rand(R, I, device="cuda", generator=g) for _ in range(3)]
st = nt._NtfState(eng, T)
for skip in (False, True):
    for _ in range(3):
        Ft2, _ = nt._one_ntf_step_dev(st, R, Ft, "hals", 2, [None] * 3, [], [False] * 3, math.inf, 0.01, skip_cost=skip)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        Ft2, _ = nt._one_ntf_step_dev(st, R, Ft, "hals", 2, [None] * 3, [], [False] * 3, math.inf, 0.01, skip_cost=skip)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"NTF step skip_cost={skip}: host enqueue {(t1-t0)*100:.3f} ms/iter, total {(t2-t0)*100:.3f} ms/iter", flush=True)
# NMF B
m, n, r = 100000, 2000, 50
X = torch.rand(m, n, device="cuda", generator=g); Ut = torch.rand(r, m, device="cuda", generator=g); V = torch.rand(r, n, device="cuda", generator=g)
ws = nm._StepBuffers(X, r)
for _ in range(3):
    nm._one_nmf_step_dev(eng, ws, X, r, Ut, V, "hals", 2, [None, None], [], [False, False], True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    nm._one_nmf_step_dev(eng, ws, X, r, Ut, V, "hals", 2, [None, None], [], [False, False], True)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"NMF step: host enqueue {(t1-t0)*100:.3f} ms/iter, total {(t2-t0)*100:.3f} ms/iter")
