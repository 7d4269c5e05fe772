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
m, n, r, iters = 6001, 300, 20, 6
X, U0, V0 = orc.synth_nmf(m, n, r, seed=4, dtype=np.float32)
lo, hi = nd.shard_rows(m, rank, world)
Xl = torch.from_numpy(X[lo:hi]).cuda()
Ut = torch.from_numpy(U0[lo:hi].T.copy()).cuda()
V = torch.from_numpy(V0).cuda()
ws = nm._StepBuffers(Xl, r)
ws.guess_u = nd.SweepGuess(first=4, max_chunk=6)
costs, sweeps = [], []
for _ in range(iters):
    Ut, V, nstat = nm._one_nmf_step_dev(eng, ws, Xl, r, Ut, V, RULE, BETA, [None, None], [], [False, False], True,
                                        group=dist.group.WORLD)
    h = ws.block.cpu(); costs.append(float(h[16])); sweeps += [int(h[8 * i + 1]) - 1 for i in range(nstat)]
# single-process reference run of the same engine on rank 0
if rank == 0:
    Xd, Ud, Vd = torch.from_numpy(X).cuda(), torch.from_numpy(U0.T.copy()).cuda(), torch.from_numpy(V0).cuda()
    ws1 = nm._StepBuffers(Xd, r); c1, s1 = [], []
    for _ in range(iters):
        Ud, Vd, nstat = nm._one_nmf_step_dev(eng, ws1, Xd, r, Ud, Vd, RULE, BETA, [None, None], [], [False, False], True)
        h = ws1.block.cpu(); c1.append(float(h[16])); s1 += [int(h[8 * i + 1]) - 1 for i in range(nstat)]
    relV = float((V - Vd).norm() / Vd.norm()); relU = float((Ut - Ud[:, lo:hi]).norm() / Ud[:, lo:hi].norm())
    print("sharded sweeps", sweeps); print("single  sweeps", s1)
    print("relV %.2e relU %.2e cost rel %.2e" % (relV, relU, abs(costs[-1] - c1[-1]) / c1[-1]))
    assert sweeps == s1 and relV < 1e-4 and relU < 1e-4 and abs(costs[-1] - c1[-1]) <= 1e-4 * c1[-1]
    print("DIST_GPU_CHECK_OK")
dist.barrier(); dist.destroy_process_group()
