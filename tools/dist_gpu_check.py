"""Two ranks sharing ONE GPU over gloo: the real HIP kernels inside the row-sharded step (dist.py) vs the
single-process run.  Launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_gpu_check.py [hals|mu] [beta]"""
import os, sys
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nnfac_oracle as orc
from nn_fac_amd import nmf as nm, dist as nd
from nn_fac_amd.engine import get_engine
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
eng = get_engine("cuda:0")
RULE = sys.argv[1] if len(sys.argv) > 1 else "hals"
BETA = float(sys.argv[2]) if len(sys.argv) > 2 else 2
BETA = int(BETA) if BETA == int(BETA) else BETA
NORM = [len(sys.argv) > 3 and sys.argv[3] == "normalize", False]      # normalise the SHARDED factor (rows of U^T across both ranks)
m, n, r, iters = 6001, 300, 20, 6
X, U0, V0 = orc.synth_nmf(m, n, r, seed=4, dtype=np.float32)
lo, hi = nd.shard_rows(m, rank, world)
Xl = torch.from_numpy(X[lo:hi]).cuda()
Ut = torch.from_numpy(U0[lo:hi].T.copy()).cuda()
V = torch.from_numpy(V0).cuda()
ws = nm._StepBuffers(Xl, r)
ws.guess_u = nd.SweepGuess(first=4, max_chunk=6, window=3)
costs, sweeps = [], []


def recorder(cs, sw):
    def retired(it, cost, s):
        cs.append(cost); sw.extend(s)
        return False
    return retired


# the product's own outer loop (status ring, cost under the next V-side solve, fused all-reduce of the V-side terms)
Ut, V = nm.run_steps(eng, ws, Xl, r, Ut, V, iters, RULE, BETA, [None, None], [], NORM, True,
                     recorder(costs, sweeps), group=dist.group.WORLD)
# single-process reference run of the same engine on rank 0
if rank == 0:
    Xd, Ud, Vd = torch.from_numpy(X).cuda(), torch.from_numpy(U0.T.copy()).cuda(), torch.from_numpy(V0).cuda()
    ws1 = nm._StepBuffers(Xd, r); c1, s1 = [], []
    Ud, Vd = nm.run_steps(eng, ws1, Xd, r, Ud, Vd, iters, RULE, BETA, [None, None], [], NORM, True,
                          recorder(c1, s1))
    relV = float((V - Vd).norm() / Vd.norm()); relU = float((Ut - Ud[:, lo:hi]).norm() / Ud[:, lo:hi].norm())
    print("sharded sweeps", sweeps); print("single  sweeps", s1)
    print("relV %.2e relU %.2e cost rel %.2e" % (relV, relU, max(abs(a - b) / b for a, b in zip(costs, c1))))
    tol = 5e-4 if NORM[0] else 1e-4      # (normalised: six iterations of 100-sweep solves, the row norms summed in another order)
    assert sweeps == s1 and relV < tol and relU < tol and len(costs) == len(c1) == iters and all(abs(a - b) <= tol * b for a, b in zip(costs, c1))
    print("DIST_GPU_CHECK_OK")
if len(sys.argv) > 4 and sys.argv[4] == "stop" and RULE == "hals":
    # the stopping test under the Gram-identity cost in a SHARDED run: `tol` between two cost differences of the run above -- both
    # ranks must take the near-threshold branch together (its cost pass is a collective) and stop where the single process stops
    d = [abs(costs[i - 1] - costs[i]) for i in range(1, len(costs))]
    tol = 0.5 * (d[2] + d[3])

    def stopper(cs):
        def retired(it, cost, s):
            cs.append(cost)
            return it > 0 and abs(cs[-2] - cs[-1]) < tol

        def revise_last(cost):
            cs[-1] = cost
        retired.revise_last = revise_last
        return retired
    Ut2 = torch.from_numpy(U0[lo:hi].T.copy()).cuda()
    ws2 = nm._StepBuffers(Xl, r)
    cs2 = []
    nm.run_steps(eng, ws2, Xl, r, Ut2, torch.from_numpy(V0).cuda(), 12, RULE, BETA, [None, None], [], NORM, True, stopper(cs2),
                 group=dist.group.WORLD, tol=tol)
    if rank == 0:
        ws3 = nm._StepBuffers(Xd, r)
        cs3 = []
        nm.run_steps(eng, ws3, Xd, r, torch.from_numpy(U0.T.copy()).cuda(), torch.from_numpy(V0).cuda(), 12, RULE, BETA,
                     [None, None], [], NORM, True, stopper(cs3), tol=tol)
        print("stop: sharded", len(cs2), "iterations, single", len(cs3), "direct cost from", ws2.direct_cost, ws3.direct_cost)
        assert len(cs2) == len(cs3) < 12 and ws2.direct_cost and ws3.direct_cost
        assert all(abs(a - b) <= 1e-4 * b for a, b in zip(cs2, cs3))
        print("DIST_GPU_STOP_OK")
dist.barrier(); dist.destroy_process_group()
