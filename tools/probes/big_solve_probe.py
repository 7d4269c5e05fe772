"""U-side solve of config E on ONE GPU (10^6 columns, rank 100: more columns than the persistent kernel holds): the library's
streaming solve against blind chunks of register-resident sweeps (dist.sharded_hals_solve with no group)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from nn_fac_amd.engine import get_engine
from nn_fac_amd import dist as _dist
eng = get_engine("cuda:0")
r = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for m in (125000, 250000, 1000000):
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.rand(r, 300, device="cuda", generator=g)
    G = (A @ A.t()).contiguous()
    cross = torch.rand(r, m, device="cuda", generator=g) * 50
    F0 = torch.rand(r, m, device="cuda", generator=g)
    for name in ("solve", "chunks"):
        F = F0.clone()
        st = torch.zeros(8, dtype=torch.float64, device="cuda")
        guess = _dist.SweepGuess()
        torch.cuda.synchronize(); t0 = time.time()
        if name == "solve":
            eng.hals_solve(cross, G, F, 100, delta=0.01, status=st)
            torch.cuda.synchronize()
            cnt = int(st[1].item())
        else:
            eps, cnt, eps0 = _dist.sharded_hals_solve(eng, cross, G, F, None, guess, budget=100, delta=0.01)
            torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"m={m:8d} r={r} {name:7s}: {dt*1e3:8.2f} ms  cnt={cnt}  checksum={float(F.double().sum()):.6e}", flush=True)
        # second call with a warmed guess
        if name == "chunks":
            F = F0.clone(); torch.cuda.synchronize(); t0 = time.time()
            eps, cnt, eps0 = _dist.sharded_hals_solve(eng, cross, G, F, None, guess, budget=100, delta=0.01)
            torch.cuda.synchronize()
            print(f"m={m:8d} r={r} chunks2: {(time.time()-t0)*1e3:8.2f} ms  cnt={cnt}  checksum={float(F.double().sum()):.6e}", flush=True)
