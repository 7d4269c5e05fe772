"""Parity at BASELINE.json's FULL sizes (configs[1..3]): the HIP path vs the CPU oracle on the same seeded inputs.

The mid-size fixtures (g5: 2000 x 500, g6: 40^3) pin the arithmetic; these runs pin the things that only exist at full size --
the one-round row tilings of X H^T / the MU kernels at m = 100000, the 96-slab split-K of W^T X, the 1563-wave persistent
U-side solve with its grid-wide stopping rule, the segmented MTTKRP over 500^3.  The oracle needs 2-6 s per iteration on
the GPU box's host cores (bench.py's cpu_baseline runs the same thing), so a couple of iterations per configuration are
affordable.  Tolerances: the stated fp32 ones of SURVEY.md 8c.
"""
import math

import numpy as np
import pytest
import torch

import nnfac_oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def config_b_data():
    return orc.synth_nmf(100000, 2000, 50, seed=0, dtype=np.float32)


def test_config_b_hals_full_size(config_b_data, built_lib):
    """configs[1]: 100000 x 2000 rank 50 HALS, 2 outer iterations (nmf.py:387-458): factors <= 5e-4, cost <= 1e-3,
    inner sweep counts of all four solves equal to the fp64 oracle's."""
    from nn_fac_amd.nmf import compute_nmf
    X, U0, V0 = config_b_data
    sw = []
    U, V, costs, _ = compute_nmf(X, 50, U0, V0, n_iter_max=2, tol=0, update_rule="hals", return_costs=True,
                                 deterministic=True, sweep_log=sw)
    swo = []
    Uo, Vo, co, _ = orc.compute_nmf(X.astype(np.float64), 50, U0.astype(np.float64), V0.astype(np.float64), n_iter_max=2,
                                    tol=0, update_rule="hals", return_costs=True, deterministic=True, sweeps=swo)
    assert sw == swo, (sw, swo)
    assert rel(U, Uo) < 5e-4 and rel(V, Vo) < 5e-4, (rel(U, Uo), rel(V, Vo))
    np.testing.assert_allclose(costs, co, rtol=1e-3)


def test_config_c_mu_kl_full_size(config_b_data, built_lib):
    """configs[2]: same data, MU beta = 1 (mu.py:84-88), 2 outer iterations: factors and cost <= 5e-5."""
    from nn_fac_amd.nmf import compute_nmf
    X, U0, V0 = config_b_data
    U, V, costs, _ = compute_nmf(X, 50, U0, V0, n_iter_max=2, tol=0, update_rule="mu", beta=1, return_costs=True,
                                 deterministic=True)
    Uo, Vo, co, _ = orc.compute_nmf(X.astype(np.float64), 50, U0.astype(np.float64), V0.astype(np.float64), n_iter_max=2,
                                    tol=0, update_rule="mu", beta=1, return_costs=True, deterministic=True)
    assert rel(U, Uo) < 5e-5 and rel(V, Vo) < 5e-5, (rel(U, Uo), rel(V, Vo))
    np.testing.assert_allclose(costs, co, rtol=5e-5)


def test_config_d_ntf_full_size(built_lib):
    """configs[3]: 500^3 rank 30 -- every mode's MTTKRP (ntf.py:448-449) vs fp64 unfold @ khatri_rao, then one
    one_ntf_step(alpha = inf) (ntf.py:422-477) vs the oracle: factors <= 5e-4, cost <= 1e-3, sweep counts equal."""
    from nn_fac_amd.engine import get_engine
    from nn_fac_amd.ntf import compute_ntf
    I, R = 500, 30
    T, F0 = orc.synth_ntf((I, I, I), R, seed=0, dtype=np.float32)
    eng = get_engine("cuda:0")
    Td = torch.from_numpy(T).cuda()
    Ft = [torch.from_numpy(f.T.copy()).cuda() for f in F0]
    T64 = T.astype(np.float64)
    F64 = [f.astype(np.float64) for f in F0]
    for mode in range(3):
        got = eng.mttkrp3(Td, Ft, mode).cpu().numpy().T
        want = orc.unfold(T64, mode) @ orc.khatri_rao(F64, skip_matrix=mode)
        assert rel(got, want) < 1e-5, (mode, rel(got, want))
    sw, swo = [], []
    F, costs, _ = compute_ntf(Td, R, [f.t() for f in Ft], n_iter_max=1, tol=0, return_costs=True, alpha=math.inf,
                              sparsity_coefficients=[None] * 3, normalize=[False] * 3, sweep_log=sw)
    Fo, co, _ = orc.compute_ntf(T64, R, F64, n_iter_max=1, tol=0, return_costs=True, alpha=math.inf,
                                sparsity_coefficients=[None] * 3, normalize=[False] * 3, sweeps=swo)
    assert sw == swo, (sw, swo)
    for k in range(3):
        assert rel(F[k].cpu().numpy(), Fo[k]) < 5e-4, (k, rel(F[k].cpu().numpy(), Fo[k]))
    np.testing.assert_allclose(costs, co, rtol=1e-3)


def test_config_e_block_sharded_equals_unsharded_on_one_gpu(built_lib):
    """configs[4]'s per-GPU block shape (rank 100, 4000 columns; 2 x 20000 rows here): the row-sharded step -- chunked
    blind sweeps + global stopping rule + the fused all-reduce of UtM | UtU -- run as TWO shards on one device, one after the
    other with the exchanges done by hand, must give the factors of the unsharded step on the concatenated rows
    (nmf.py:387-458).  Exercises the rank-100 sweep kernels with snapshots, which no BASELINE-sized run on one
    GPU reaches otherwise."""
    from nn_fac_amd.engine import get_engine
    from nn_fac_amd import nmf as nmf_mod
    m, n, r = 40000, 4000, 100
    X, U0, V0 = orc.synth_nmf(m, n, r, seed=5, dtype=np.float32)
    eng = get_engine("cuda:0")
    Xd = torch.from_numpy(X).cuda()
    Ut0 = torch.from_numpy(U0.T.copy()).cuda()
    Vd = torch.from_numpy(V0).cuda()
    ws = nmf_mod._StepBuffers(Xd, r)
    Ut1, V1, nstat = nmf_mod._one_nmf_step_dev(eng, ws, Xd, r, Ut0, Vd, "hals", 2, [None, None], [], [False, False], True)
    host = ws.block.cpu()
    want_sweeps = [int(host[8 * i + 1]) - 1 for i in range(nstat)]
    # U side through the chunked protocol on the whole block (one rank, no group): same factor bit for bit
    from nn_fac_amd import dist as nd
    G = eng.gram(Vd)
    VMt = eng.xht(Xd, Vd)
    F = Ut0.clone()
    eps, cnt, eps0 = nd.sharded_hals_solve(eng, VMt, G, F, None, nd.SweepGuess(first=6, max_chunk=16, window=4))
    assert cnt - 1 == want_sweeps[0]
    assert torch.equal(F, Ut1)
    # and against the oracle
    Uo, Vo, co = orc.one_nmf_step(X.astype(np.float64), r, U0.astype(np.float64), V0.astype(np.float64), None, "hals", 2,
                                  [None, None], [], [False, False], True)
    assert rel(Ut1.t().cpu().numpy(), Uo) < 5e-4 and rel(V1.cpu().numpy(), Vo) < 5e-4
    assert abs(float(host[16]) - co) <= 1e-3 * co


def test_bench_single_rank_rccl_smoke(built_lib):
    """bench.py under torchrun with ONE rank and a real RCCL process group (backend "nccl" on ROCm): init, barrier, the
    max-over-ranks all-reduce of the timing and destroy all run on the GPU -- the N-rank launch path of the driver's
    scaling runs, as far as a one-GPU box can take it."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, NNF_BENCH_INIT_PG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--shape", "20000,500,20", "--no-cpu", "--no-extra", "--no-fixed"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and "nccl" in out["config"]["parallelism"] and out["value"] > 0
    assert out["roofline"]["kernel"].startswith("nnf_xty_kernel") and out["roofline"]["launch_ms"] > 0


@pytest.mark.parametrize("cfg,switches", [("B", False), ("B", True), ("C", True), ("D", False)])
def test_sharded_protocol_on_a_one_rank_rccl_group(built_lib, cfg, switches):
    """The row-sharded step over RCCL with the device-side stopping decision, the overlapped cost (both opt-in:
    NNF_SHARDED_ASYNC=1, NNF_SHARDED_OVERLAP=1 -- off by default until an N > 1 run on real GPUs has exercised them) and (MU)
    the fused KL cost + scalar all-reduce switched on -- on the one rank a one-GPU box offers (NNF_FORCE_SHARDED=1): same data, same number of iterations as the unsharded run of the same bench
    command; HALS iterates and costs agree to the tolerance of two differently ordered fp32 sums of the stopping scalar, the MU
    line to rounding.  Config D: the leading-mode-sharded NTF step (mode-0 solve through the chunked protocol, the other modes' MTTKRP
    + mode-0 Gram in one all-reduce, the cost from the replicated operands of the last mode -- no collective)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(forced):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, NNF_BENCH_INIT_PG="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NNF_BENCH_DEBUG="1")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NNF_SHARDED_ASYNC", "NNF_SHARDED_OVERLAP"):
            env.pop(k, None)
        if forced:
            env.update(NNF_BENCH_FORCE_SHARDED="1", NNF_FORCE_SHARDED="1")
            if switches:   # both opt-in switches on: the paths with the most moving parts; off: what the driver's N > 1 runs use
                env.update(NNF_SHARDED_ASYNC="1", NNF_SHARDED_OVERLAP="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py"), "--config", cfg, "--gpus", "1", "--steps", "12",
               "--warmup", "2", "--shape", "160,160,12" if cfg == "D" else "30000,600,18", "--no-cpu", "--no-extra", "--no-fixed",
               "--no-kernels"]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0]), p.stderr

    plain, _ = run(False)
    shard, err = run(True)
    assert shard["n_gpus"] == 1 and "nccl" in shard["config"]["parallelism"]
    a, b = plain["config"]["final_cost"], shard["config"]["final_cost"]
    assert abs(a - b) <= (1e-5 if cfg == "C" else 1e-3) * abs(a), (a, b)
    if cfg == "D":
        assert plain["config"]["inner_sweeps_per_step_last"] == shard["config"]["inner_sweeps_per_step_last"]
    if cfg == "B":
        assert plain["config"]["inner_sweeps_per_step_last"] == shard["config"]["inner_sweeps_per_step_last"]
        assert "sharded U-side protocol" in err          # the protocol really ran (bench.py's debug line)
        sp = shard["config"]["sharded_protocol"]
        assert sp["u_side_stopping_decision"].startswith("device" if switches else "host-synchronous") and not sp["fell_back_to_chunked_solves"]


def test_config_e_full_size_on_one_gpu(built_lib):
    """configs[4] at its FULL size on one GPU (10^6 x 4000, rank 100: X is 16 GB, never on the host; bench.py's generator):
    one HALS iteration through the product's step, checked by size-independent properties -- the oracle would need ~50 GB and
    minutes here.  (a) V X^T and U^T X against torch products of row chunks (the 64-bit row offsets, the multi-round tilings
    and the 1e6-column strided U-side solve only exist at this size); (b) the cost against a chunked fp64 residual; (c) the
    U-side solve, whose columns are independent, against the SAME number of sweeps run on one 50000-column slice alone
    (resident kernel instead of the strided one); (d) the V-side solve against a re-solve from the same operands (bitwise)."""
    import bench
    from nn_fac_amd.engine import get_engine, ST_CNT, ST_ERR
    from nn_fac_amd import nmf as nmf_mod
    m, n, r = 1000000, 4000, 100
    dev = torch.device("cuda:0")
    eng = get_engine(dev)
    parts = [bench.synth_nmf_block_device(m // 8, n, r, b, 977, dev, torch) for b in range(8)]
    X = torch.cat([p[0] for p in parts])
    Ut0 = torch.cat([p[1] for p in parts]).t().contiguous()
    del parts
    V0 = torch.rand(r, n, device=dev, generator=torch.Generator(device=dev).manual_seed(4242))
    ws = nmf_mod._StepBuffers(X, r)
    Ut1, V1, nstat = nmf_mod._one_nmf_step_dev(eng, ws, X, r, Ut0, V0, "hals", 2, [None, None], [], [False, False], True)
    host = ws.block.cpu()
    assert nstat == 2 and int(host[ST_ERR]) == 0 and int(host[8 + ST_ERR]) == 0
    su, sv = int(host[ST_CNT]) - 1, int(host[8 + ST_CNT]) - 1
    assert 2 <= su <= 100 and 2 <= sv <= 100

    def relerr(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())
    # (a) the cross products, on three row chunks spread over the matrix (incl. the last rows)
    VMt = eng.xht(X, V0)
    UtM = eng.xty(X, Ut1)
    acc = torch.zeros(r, n, dtype=torch.float64, device=dev)
    for lo in range(0, m, 125000):
        acc += Ut1[:, lo:lo + 125000].double() @ X[lo:lo + 125000].double()
    assert relerr(UtM, acc) < 1e-5
    for lo in (0, 437500, 999000):
        hi = min(m, lo + 1000)
        assert relerr(VMt[:, lo:hi], V0.double() @ X[lo:hi].double().t()) < 1e-5
    # (b) the cost of the step
    want = 0.0
    for lo in range(0, m, 125000):
        want += float(((X[lo:lo + 125000].double() - Ut1[:, lo:lo + 125000].double().t() @ V1.double()) ** 2).sum())
    assert abs(float(host[16]) - want) <= 1e-5 * want, (float(host[16]), want)
    # (c) columns of the U-side solve are independent given the sweep count
    G = eng.gram(V0)
    lo, hi = 300000, 350000
    F = Ut0[:, lo:hi].contiguous()
    eng.hals_sweeps(VMt[:, lo:hi].contiguous(), G, F, su)
    assert relerr(F, Ut1[:, lo:hi]) < 1e-5
    # (d) the replicated V-side solve is deterministic
    G2 = eng.gram(Ut1)
    V2 = V0.clone()
    st = eng.hals_solve(UtM, G2, V2, 100, delta=0.01).cpu()
    assert int(st[ST_CNT]) - 1 == sv and torch.equal(V2, V1)


def _identity_vs_direct(X, Ut0, V0, r, iters, monkeypatch, tol_rel):
    """`iters` HALS iterations of the product's loop, once with the Gram-identity cost and once with NNF_COST=direct: returns the
    two cost lists, the factors of the identity run and the kind of every identity-run cost."""
    from nn_fac_amd.engine import get_engine
    from nn_fac_amd import nmf as nmf_mod
    eng = get_engine(X.device)
    out = {}
    for kind in ("identity", "direct"):
        if kind == "direct":
            monkeypatch.setenv("NNF_COST", "direct")
        else:
            monkeypatch.delenv("NNF_COST", raising=False)
        ws = nmf_mod._StepBuffers(X, r)
        costs = []

        def retired(it, cost, sw):
            costs.append(float(cost))
            return False
        retired.revise_last = lambda c: costs.__setitem__(-1, float(c))
        Ut, V = nmf_mod.run_steps(eng, ws, X, r, Ut0.clone(), V0.clone(), iters, "hals", 2, [None, None], [], [False, False], True,
                                  retired)
        out[kind] = (costs, Ut, V, ws)
    monkeypatch.delenv("NNF_COST", raising=False)
    return out


def _chunked_fp64_cost(X, Ut, V, step=50000):
    want = 0.0
    for lo in range(0, X.shape[0], step):
        want += float(((X[lo:lo + step].double() - Ut[:, lo:lo + step].double().t() @ V.double()) ** 2).sum())
    return want


def test_identity_cost_tracks_the_direct_cost_over_a_whole_run_at_config_b(config_b_data, built_lib, monkeypatch):
    """configs[1], 25 iterations (the driver's default run): the Gram-identity cost of EVERY iteration against the streaming
    kernel's (NNF_COST=direct: same factors, the cost does not feed back into them) <= 1e-4, and the last one against a chunked
    fp64 residual of the final factors.  The identity's error is relative to ||X||^2, so it grows as the fit improves: two
    iterations (test_config_b_hals_full_size) do not show it."""
    X, U0, V0 = config_b_data
    Xd = torch.from_numpy(X).cuda()
    Ut0 = torch.from_numpy(U0.T.copy()).cuda()
    Vd = torch.from_numpy(V0).cuda()
    res = _identity_vs_direct(Xd, Ut0, Vd, 50, 25, monkeypatch, 1e-4)
    ci, cd = res["identity"][0], res["direct"][0]
    assert len(ci) == len(cd) == 25
    assert torch.equal(res["identity"][1], res["direct"][1]) and torch.equal(res["identity"][2], res["direct"][2])
    worst = max(abs(a - b) / b for a, b in zip(ci, cd))
    assert worst <= 1e-4, (worst, ci[-3:], cd[-3:])
    want = _chunked_fp64_cost(Xd, res["identity"][1], res["identity"][2])
    assert abs(ci[-1] - want) <= 1e-4 * want and abs(cd[-1] - want) <= 1e-6 * want, (ci[-1], cd[-1], want)
    assert not res["identity"][3].direct_cost          # the guard did not have to fall back on this data


@pytest.mark.parametrize("rows", [125000, 1000000])
def test_identity_cost_at_rank_100_long_accumulations(rows, built_lib, monkeypatch):
    """configs[4]'s shapes -- the per-rank block (125000 x 4000, rank 100) and the whole problem on one device (1e6 rows) --
    12 iterations.  The rounding of a U^T X entry depends on the rows one workgroup of the W^T X kernel sums in fp32 (9.5e-7
    rms with a -2.2e-7 MEAN at 62500 rows per workgroup, 6e-8 ... 1e-7 with a mean below 1e-9 at the 2048 the launch plan now
    allows: tools/probes/accum_error_probe.py), so the error estimate of the identity cost is calibrated per run
    (Engine.cross_rounding / gram_rounding).  Every identity cost that the loop kept must be within 5e-4 of the direct cost; an
    iterate the guard flags switches the run to the direct cost (then the costs are the direct run's bit for bit); and the
    calibration must report short chains (a plan that went back to long ones would show here first)."""
    import bench
    from nn_fac_amd.engine import get_engine
    m, n, r = rows, 4000, 100
    dev = torch.device("cuda:0")
    eng = get_engine(dev)
    nb = rows // 125000
    parts = [bench.synth_nmf_block_device(125000, n, r, b, 977, dev, torch) for b in range(nb)]
    X = torch.cat([p[0] for p in parts]) if nb > 1 else parts[0][0]
    Ut0 = (torch.cat([p[1] for p in parts]) if nb > 1 else parts[0][1]).t().contiguous()
    del parts
    V0 = torch.rand(r, n, device=dev, generator=torch.Generator(device=dev).manual_seed(4242))
    sa, ba = eng.cross_rounding(X, Ut0)
    assert 2e-8 < sa < 2e-7 and ba < 2e-9, (sa, ba)
    assert eng.gram_rounding(Ut0) < 2e-8
    res = _identity_vs_direct(X, Ut0, V0, r, 12, monkeypatch, 5e-4)
    ci, cd = res["identity"][0], res["direct"][0]
    assert torch.equal(res["identity"][1], res["direct"][1]) and torch.equal(res["identity"][2], res["direct"][2])
    worst = max(abs(a - b) / b for a, b in zip(ci, cd))
    assert worst <= 5e-4, (worst, ci, cd)
    want = _chunked_fp64_cost(X, res["direct"][1], res["direct"][2], step=25000)
    assert abs(cd[-1] - want) <= 1e-5 * want
    if res["identity"][3].direct_cost:                       # flagged on the way: from there on the direct run's costs
        assert ci[-1] == cd[-1]
