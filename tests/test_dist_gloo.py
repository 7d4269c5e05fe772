"""Row-sharded NMF step over torch.distributed, world_size 2, gloo on CPU.

The HIP engine cannot run here, so the per-rank compute is a TEST DOUBLE backed by the CPU oracle (NumPy, fp64):
what is under test is the host logic of nn_fac_amd/dist.py + nn_fac_amd/nmf.py -- the row partition, the all-reduce
of the Gram / cross terms / cost, and the chunk-and-replay protocol for the global HALS stopping rule -- which must
reproduce the unsharded oracle exactly (same sweep counts, same factors to fp64 round-off).
"""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nnfac_oracle as orc


from engine_double import OracleEngine  # noqa: E402


def _worker(rank, nranks, port, m, n, r, iters, sparsity, q, rule="hals", beta=2, guess=(3, 5, 2), deterministic=True,
            normalize=(False, False)):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=nranks)
    try:
        from nn_fac_amd import nmf as nmf_mod, dist as nd
        X, U0, V0 = orc.synth_nmf(m, n, r, seed=1, dtype=np.float64)
        lo, hi = nd.shard_rows(m, rank, nranks)
        Xl = torch.from_numpy(X[lo:hi].copy())
        Ut = torch.from_numpy(U0[lo:hi].T.copy())
        V = torch.from_numpy(V0.copy())
        # the product's own buffers and outer loop (status ring, fused all-reduce of the V-side terms), engine double below
        eng, ws = OracleEngine(), nmf_mod._StepBuffers(Xl, r, dtype=torch.float64)
        # (3, 5, 2): small on purpose -- the device-side protocol always misses (chunk shorter than the solve) and every
        # iteration is redone through the synchronous one: continue, exact stop, snapshot, replay.  (16, 104, 8): the
        # defaults -- misses while the guess settles, then hits
        ws.guess_u = nd.SweepGuess(first=guess[0], max_chunk=guess[1], window=guess[2])
        ws.async_sharded, ws.async_ready = True, True       # (opt-in in the product: NNF_SHARDED_ASYNC=1)
        costs, sweeps = [], []

        def retired(it, cost, sw):
            costs.append(cost)
            sweeps.extend(sw)
            return False

        Ut, V = nmf_mod.run_steps(eng, ws, Xl, r, Ut, V, iters, rule, beta, sparsity, [], list(normalize), deterministic, retired,
                                  group=dist.group.WORLD)
        q.put((rank, lo, hi, Ut.numpy().T.copy(), V.numpy().copy(), costs, sweeps, (ws.async_hits, ws.async_misses)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("guess", [(3, 5, 2), (16, 104, 8)])
@pytest.mark.parametrize("sparsity", [[None, None], [0.05, 0.02]])
def test_row_sharded_step_equals_unsharded_oracle(sparsity, guess):
    m, n, r, iters, nranks = 301, 40, 6, 8, 2          # odd m: unequal shards
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(k, nranks, port, m, n, r, iters, sparsity, q, "hals", 2, guess))
             for k in range(nranks)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(nranks))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, U0, V0 = orc.synth_nmf(m, n, r, seed=1, dtype=np.float64)
    sw = []
    U, V, costs, _ = orc.compute_nmf(X, r, U0, V0, n_iter_max=iters, tol=0, update_rule="hals",
                                     sparsity_coefficients=list(sparsity), return_costs=True, deterministic=True,
                                     sweeps=sw)
    Ucat = np.concatenate([x[3] for x in res], axis=0)
    np.testing.assert_allclose(Ucat, U, rtol=1e-9, atol=1e-12)
    for rank, lo, hi, Ul, Vl, cl, sl, (hits, misses) in res:
        np.testing.assert_allclose(Vl, V, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(cl, costs, rtol=1e-9)
        assert sl == sw                                   # identical inner sweep counts on every rank
        assert hits + misses >= 1                         # the device-side protocol was exercised
        if guess[1] >= 100:
            assert hits >= 1, (hits, misses, sw)          # ... and once the guess has settled it hits
        else:
            assert hits == 0 or max(sw[0::2]) <= 4        # a 5-sweep chunk cannot contain a longer solve's stop
    assert np.array_equal(res[0][4], res[1][4])           # replicated V bitwise identical across ranks


@pytest.mark.parametrize("sparsity", [[None, None], [0.03, None]])
def test_row_sharded_step_with_a_normalised_sharded_factor(sparsity):
    """nmf(normalize=[True, .]) over two ranks: the norm of a row of U^T (nnls.py:179-185) runs over the columns of BOTH ranks,
    once per row update -- dist.sharded_hals_solve_rownorm walks the rows with one all-reduce each.  Factors, costs and sweep
    counts of the unsharded oracle (round 2 raised NotImplementedError here)."""
    m, n, r, iters, nranks = 151, 30, 5, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(k, nranks, port, m, n, r, iters, sparsity, q, "hals", 2, (3, 5, 2), True, (True, False)))
             for k in range(nranks)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(nranks))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, U0, V0 = orc.synth_nmf(m, n, r, seed=1, dtype=np.float64)
    sw = []
    U, V, costs, _ = orc.compute_nmf(X, r, U0, V0, n_iter_max=iters, tol=0, update_rule="hals", sparsity_coefficients=list(sparsity),
                                     normalize=[True, False], return_costs=True, deterministic=True, sweeps=sw)
    np.testing.assert_allclose(np.concatenate([x[3] for x in res], axis=0), U, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(U, axis=0), 1.0, rtol=1e-12)       # (the columns of U are what got normalised)
    for rank, lo, hi, Ul, Vl, cl, sl, _ in res:
        np.testing.assert_allclose(Vl, V, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(cl, costs, rtol=1e-9)
        assert sl == sw


@pytest.mark.parametrize("beta", [1, 2, 0.5, 3])
def test_row_sharded_mu_step_equals_unsharded_oracle(beta):
    """MU: left update local, right update from all-reduced numerator / denominator, cost all-reduced."""
    m, n, r, iters, nranks = 203, 30, 5, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(k, nranks, port, m, n, r, iters, [None, None], q, "mu", beta))
             for k in range(nranks)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(nranks))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, U0, V0 = orc.synth_nmf(m, n, r, seed=1, dtype=np.float64)
    U, V, costs, _ = orc.compute_nmf(X, r, U0, V0, n_iter_max=iters, tol=0, update_rule="mu", beta=beta,
                                     return_costs=True, deterministic=True)
    np.testing.assert_allclose(np.concatenate([x[3] for x in res], axis=0), U, rtol=1e-9, atol=1e-12)
    for rank, lo, hi, Ul, Vl, cl, sl, _ in res:
        np.testing.assert_allclose(Vl, V, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(cl, costs, rtol=1e-9)


def test_shard_rows_partition():
    from nn_fac_amd.dist import shard_rows
    for m, k in ((10, 3), (100000, 8), (7, 8), (301, 2)):
        edges = [shard_rows(m, i, k) for i in range(k)]
        assert edges[0][0] == 0 and edges[-1][1] == m
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        sizes = [hi - lo for lo, hi in edges]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("first,max_chunk,window", [(40, 64, 4), (3, 5, 2), (6, 8, 8), (100, 104, 8), (2, 3, 1)])
@pytest.mark.parametrize("delta,budget", [(0.01, 100), (0.3, 100), (0.0, 7), (0.01, 1)])
def test_chunked_solve_protocol_single_rank(first, max_chunk, window, delta, budget):
    """dist.sharded_hals_solve on one rank (no process group): whatever the chunk length, the snapshot window and the
    position of the stopping sweep -- in the window (snapshot), before it (copy + re-run), at the end of a chunk, in a later
    chunk, at the sweep budget -- the result is the straight solve of nnls.py:156-196, bit for bit in the double."""
    from nn_fac_amd import dist as nd
    rng = np.random.RandomState(first * 7 + window)
    r, n = 6, 50
    U = rng.rand(80, r)
    M = U @ rng.rand(r, n) + 1e-2 * rng.rand(80, n)
    UtU, UtM, V0 = U.T @ U, U.T @ M, rng.rand(r, n)
    want, eps, cnt, _ = orc.hals_nnls_acc(UtM, UtU, V0, maxiter=budget, alpha=math.inf, delta=delta)
    calls = []

    class Counting(OracleEngine):
        def hals_sweeps(self, UtM, UtU, V, nsweeps, **kw):
            calls.append((nsweeps - kw.get("snap_first", 0), kw.get("snapshots") is not None))
            return super().hals_sweeps(UtM, UtU, V, nsweeps, **kw)

    F = torch.from_numpy(V0.copy())
    guess = nd.SweepGuess(first=first, max_chunk=max_chunk, window=window)
    e2, c2, _ = nd.sharded_hals_solve(Counting(), torch.from_numpy(UtM), torch.from_numpy(UtU), F, None, guess,
                                      budget=budget, delta=delta)
    assert c2 == cnt and np.array_equal(F.numpy(), want) and e2 == eps
    assert all(ns <= window for ns, snapped in calls if snapped)          # a chunk is one launch; only its window takes snapshots
    # a second call starts from the remembered count: one chunk, stop inside its window, no re-run
    calls.clear()
    F = torch.from_numpy(V0.copy())
    nd.sharded_hals_solve(Counting(), torch.from_numpy(UtM), torch.from_numpy(UtU), F, None, guess, budget=budget, delta=delta)
    assert np.array_equal(F.numpy(), want)
    if cnt - 1 + 4 <= max_chunk and cnt - 1 >= 8:
        assert len(calls) <= 2, calls


@pytest.mark.parametrize("first,window", [(3, 2), (30, 4), (100, 8)])
def test_chunked_solve_over_column_blocks(first, window):
    """dist.column_blocks / _blind_sweeps: an engine whose resident sweep kernel holds fewer columns than the factor has (config E
    on one, two or four devices) runs the blind chunks block by block, adds the blocks' per-sweep sums in block order, restores
    from per-block snapshots -- and must return what the one-block run returns, in both forms of the stopping decision."""
    from nn_fac_amd import dist as nd
    rng = np.random.RandomState(first)
    r, n = 5, 700
    U = rng.rand(60, r)
    M = U @ rng.rand(r, n) + 1e-2 * rng.rand(60, n)
    UtU, UtM, V0 = U.T @ U, U.T @ M, rng.rand(r, n)
    want, eps, cnt, _ = orc.hals_nnls_acc(UtM, UtU, V0, maxiter=100, alpha=math.inf, delta=0.01)
    seen = []

    class Small(OracleEngine):
        def hals_resident_columns(self, rank):
            return 256

        def hals_sweeps(self, UtM, UtU, V, nsweeps, **kw):
            seen.append(int(V.shape[1]))
            return super().hals_sweeps(UtM, UtU, V, nsweeps, **kw)

    eng = Small()
    assert nd.column_blocks(eng, torch.from_numpy(V0)) == [(0, 256), (256, 512), (512, 700)]
    F = torch.from_numpy(V0.copy())
    e2, c2, _ = nd.sharded_hals_solve(eng, torch.from_numpy(UtM), torch.from_numpy(UtU), F, None,
                                      nd.SweepGuess(first=first, max_chunk=104, window=window), budget=100, delta=0.01)
    assert c2 == cnt and max(seen) == 256 and np.allclose(F.numpy(), want, rtol=1e-12, atol=0) and abs(e2 - eps) <= 1e-12 * eps
    if cnt - 1 <= first and cnt - 1 > first - window:      # the device-side form needs the stop inside its one chunk's window
        F = torch.from_numpy(V0.copy())
        status = torch.zeros(8, dtype=torch.float64)
        nd.sharded_hals_solve_async(eng, torch.from_numpy(UtM), torch.from_numpy(UtU), F, None,
                                    nd.SweepGuess(first=first, max_chunk=104, window=window), status, budget=100, delta=0.01)
        assert int(status[1]) == cnt and int(status[3]) == 0 and np.allclose(F.numpy(), want, rtol=1e-12, atol=0)


def _ntf_worker(rank, nranks, port, shape, R, iters, sparsity, q, async_guess=None):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=nranks)
    try:
        from nn_fac_amd import ntf as ntf_mod, dist as nd
        T, F0 = orc.synth_ntf(shape, R, seed=3, dtype=np.float64)
        lo, hi = nd.shard_rows(shape[0], rank, nranks)
        st = ntf_mod._NtfState(OracleEngine(), torch.from_numpy(T[lo:hi].copy()), group=dist.group.WORLD)
        st.guess0 = nd.SweepGuess(first=3, max_chunk=5, window=2)
        if async_guess is not None:      # the device-side decision from the first iteration on (over gloo it is off by default)
            st.async_sharded, st.async_ready = True, True
            st.guess0 = nd.SweepGuess(first=async_guess[0], max_chunk=async_guess[1], window=async_guess[2])
        Ft = [torch.from_numpy(F0[0][lo:hi].T.copy())] + [torch.from_numpy(f.T.copy()) for f in F0[1:]]
        costs, sweeps = [], []

        def retired(it, cost, sw):
            costs.append(cost)
            sweeps.extend(sw)
            return False

        Ft = ntf_mod.run_ntf_steps(st, R, Ft, iters, "hals", 2, list(sparsity), [], [False] * 3, math.inf, 0.01, retired)
        q.put((rank, lo, hi, [f.numpy().T.copy() for f in Ft], costs, sweeps, (st.async_hits, st.async_misses)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sparsity,async_guess", [([None, None, None], None), ([0.03, 0.02, None], None),
                                                  ([None, None, None], (104, 104, 104)),     # one blind chunk holds every stop
                                                  ([0.03, 0.02, None], (2, 3, 2))])          # chunks too short: every guess missed
def test_leading_mode_sharded_ntf_equals_unsharded_oracle(sparsity, async_guess):
    """NTF with the leading mode sharded: mode-0 update local + global stopping scalar, the other modes from
    the all-reduced MTTKRP output and mode-0 Gram, cost all-reduced -- must reproduce the unsharded oracle (ntf.py:422-477).
    `async_guess`: the mode-0 solve with the device-side stopping decision (hits) and its rewind to the host-synchronous
    protocol (misses), as in the NMF step."""
    shape, R, iters, nranks = (23, 9, 7), 4, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ntf_worker, args=(k, nranks, port, shape, R, iters, sparsity, q, async_guess))
             for k in range(nranks)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(nranks))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T, F0 = orc.synth_ntf(shape, R, seed=3, dtype=np.float64)
    sw = []
    F, costs, _ = orc.compute_ntf(T, R, F0, n_iter_max=iters, tol=0, sparsity_coefficients=list(sparsity),
                                  normalize=[False] * 3, return_costs=True, alpha=math.inf, sweeps=sw)
    np.testing.assert_allclose(np.concatenate([x[3][0] for x in res], axis=0), F[0], rtol=1e-9, atol=1e-12)
    for rank, lo, hi, Fl, cl, sl, (hits, misses) in res:
        for k in (1, 2):
            np.testing.assert_allclose(Fl[k], F[k], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(cl, costs, rtol=1e-7, atol=1e-13)
        assert sl == sw
        if async_guess == (104, 104, 104):
            assert hits >= 2 and misses == 0      # (the settling rule takes one iteration out: no previous count yet)
        elif async_guess is not None:
            assert misses >= 1


@pytest.mark.parametrize("config,shape", [("B", "301,40,6"), ("C", "203,30,5"), ("D", "14,14,3")])
def test_bench_spawns_its_own_ranks(config, shape):
    """`python bench.py --gpus 2` outside torchrun starts two ranks itself and rank 0 reports n_gpus = 2 (the launch path the
    driver's --gpus N runs use; here over gloo with the CPU engine double, tiny shapes)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NNF_BENCH_BACKEND="gloo", NNF_BENCH_ENGINE="engine_double:OracleEngine",
               PYTHONPATH=os.pathsep.join([os.path.join(root, "tests"), root, os.environ.get("PYTHONPATH", "")]))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--config", config, "--shape", shape, "--no-cpu"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["parallelism"].startswith("row-sharded x2")


def test_run_steps_falls_back_to_chunked_solves_after_a_timeout():
    """nn_fac_amd.nmf.run_steps: a persistent HALS solve that reports a time-out (status word 1: its workgroups were not all
    resident, e.g. a GPU shared with another process) must not kill the factorisation -- everything in flight is dropped, the
    loop resumes from the last clean iteration with chunked fixed-count solves, and the run ends with the factors, costs and
    sweep counts of an undisturbed one (engine double: the host logic is what is under test)."""
    import warnings
    from nn_fac_amd import nmf as nmf_mod
    X, U0, V0 = orc.synth_nmf(120, 30, 5, seed=4, dtype=np.float64)
    iters = 6

    class Flaky(OracleEngine):
        calls = 0

        def hals_solve(self, *a, **kw):
            Flaky.calls += 1
            st = super().hals_solve(*a, **kw)
            if Flaky.calls == 5:                 # V-side solve of the third iteration
                st[3] = 1.0
            return st

    def run(eng):
        Xt = torch.from_numpy(X)
        ws = nmf_mod._StepBuffers(Xt, 5, dtype=torch.float64)
        ws.guess_u = nmf_mod._dist.SweepGuess(first=3, max_chunk=5, window=2)
        ws.guess_v = nmf_mod._dist.SweepGuess(first=3, max_chunk=5, window=2)
        costs, sweeps = [], []

        def retired(it, cost, sw):
            costs.append((it, cost))
            sweeps.append(sw)
            return False
        Ut, V = nmf_mod.run_steps(eng, ws, Xt, 5, torch.from_numpy(U0.T.copy()), torch.from_numpy(V0.copy()), iters, "hals", 2,
                                  [None, None], [], [False, False], True, retired)
        return Ut.numpy(), V.numpy(), costs, sweeps, ws

    Ut0, Vr0, c0, s0, _ = run(OracleEngine())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        Ut1, Vr1, c1, s1, ws1 = run(Flaky())
    assert ws1.safe_solve and any("timed out" in str(x.message) for x in w)
    assert [i for i, _ in c1] == list(range(iters))          # every iteration retired exactly once, in order
    np.testing.assert_allclose(Ut1, Ut0, rtol=1e-12)
    np.testing.assert_allclose(Vr1, Vr0, rtol=1e-12)
    np.testing.assert_allclose([c for _, c in c1], [c for _, c in c0], rtol=1e-12)
    assert s1 == s0


def _flaky_worker(rank, nranks, port, m, n, r, iters, flaky_rank, q):
    """One rank of a row-sharded run whose replicated V-side solve reports a time-out on `flaky_rank` ONLY."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=nranks)
    try:
        import warnings
        from nn_fac_amd import nmf as nmf_mod, dist as nd
        X, U0, V0 = orc.synth_nmf(m, n, r, seed=4, dtype=np.float64)
        lo, hi = nd.shard_rows(m, rank, nranks)
        Xl = torch.from_numpy(X[lo:hi].copy())

        class Flaky(OracleEngine):
            calls = 0

            def hals_solve(self, *a, **kw):
                Flaky.calls += 1
                st = super().hals_solve(*a, **kw)
                if Flaky.calls == 3 and rank == flaky_rank:      # the V-side solve of the third iteration, on one rank
                    st[3] = 1.0
                return st

        eng, ws = Flaky(), nmf_mod._StepBuffers(Xl, r, dtype=torch.float64)
        ws.guess_u = nd.SweepGuess(first=3, max_chunk=5, window=2)
        ws.guess_v = nd.SweepGuess(first=3, max_chunk=5, window=2)
        costs, sweeps = [], []

        def retired(it, cost, sw):
            costs.append((it, cost))
            sweeps.extend(sw)
            return False

        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            Ut, V = nmf_mod.run_steps(eng, ws, Xl, r, torch.from_numpy(U0[lo:hi].T.copy()), torch.from_numpy(V0.copy()), iters,
                                      "hals", 2, [None, None], [], [False, False], True, retired, group=dist.group.WORLD)
        q.put((rank, Ut.numpy().T.copy(), V.numpy().copy(), costs, sweeps, bool(ws.safe_solve),
               any("timed out" in str(x.message) for x in w)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("flaky_rank", [0, 1])
def test_row_sharded_timeout_fall_back_is_taken_by_every_rank(flaky_rank):
    """A persistent solve that times out is a rank-local event (ONE rank's replicated V-side solve found the chip shared).
    The error word travels with the cost's all-reduce (dist.allreduce_cost_), so BOTH ranks drop what is in flight, rewind
    to the same iteration and switch to chunked solves together -- the collectives keep matching -- and the run ends with
    the factors, costs and sweep counts of the undisturbed unsharded oracle."""
    m, n, r, iters, nranks = 121, 30, 5, 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flaky_worker, args=(k, nranks, port, m, n, r, iters, flaky_rank, q)) for k in range(nranks)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(nranks))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, U0, V0 = orc.synth_nmf(m, n, r, seed=4, dtype=np.float64)
    sw = []
    U, V, costs, _ = orc.compute_nmf(X, r, U0, V0, n_iter_max=iters, tol=0, update_rule="hals", return_costs=True,
                                     deterministic=True, sweeps=sw)
    np.testing.assert_allclose(np.concatenate([x[1] for x in res], axis=0), U, rtol=1e-9, atol=1e-12)
    for rank, Ul, Vl, cl, sl, safe, warned in res:
        assert safe and warned, (rank, safe, warned)       # the rank that did NOT time out fell back as well
        assert [i for i, _ in cl] == list(range(iters))     # every iteration retired exactly once, in order
        np.testing.assert_allclose(Vl, V, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose([c for _, c in cl], costs, rtol=1e-9)
        assert sl == sw
    assert np.array_equal(res[0][2], res[1][2])


def test_sharded_random_init_reproduces_the_reference_stream():
    """dist.sharded_random_init(exact_stream=True): the blocks of all ranks concatenated are the unsharded start values of the
    reference (initialize_factors.py:40-46: np.random.seed(seed); rand(m, r); rand(r, n)) = the oracle's nmf_random_init;
    the device-style default gives every rank the same V_0 and distinct row blocks."""
    from nn_fac_amd import dist as nd
    m, n, r, seed = 103, 17, 5, 3
    U, V = orc.nmf_random_init((m, n), r, seed)

    class FakeGroup:               # shard_rows / world are pure functions of (rank, nranks): emulate three ranks in-process
        pass
    blocks = []
    for k in range(3):
        lo, hi = nd.shard_rows(m, k, 3)
        orig_world, orig_rank = nd.world, dist.get_rank
        nd.world, dist.get_rank = (lambda g: 3), (lambda g=None: k)
        try:
            U0, V0, (a, b) = nd.sharded_random_init(m, n, r, FakeGroup(), seed=seed, exact_stream=True)
            Ud, Vd, _ = nd.sharded_random_init(m, n, r, FakeGroup(), seed=seed)
        finally:
            nd.world, dist.get_rank = orig_world, orig_rank
        assert (a, b) == (lo, hi) and np.array_equal(V0.numpy(), V)
        blocks.append((U0.numpy(), Ud, Vd))
    assert np.array_equal(np.concatenate([b[0] for b in blocks]), U)
    assert all(torch.equal(blocks[0][2], b[2]) for b in blocks)                 # same V_0 on every rank
    assert not torch.equal(blocks[0][1][:10], blocks[1][1][:10])                # different row blocks


def test_row_sharded_wall_clock_rule_agrees_across_ranks():
    """deterministic=False (the reference's default: cnt <= 1 + alpha * atime / btime, nnls.py:156,190-194): the sweep budget
    comes from the wall clock, which differs from rank to rank -- rank 0's figure is broadcast, so that the ranks run the same
    sweeps and the replicated V stays bitwise identical.  Results are time dependent by design: only the agreement and the
    descent are checked."""
    m, n, r, iters, nranks = 211, 30, 5, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(k, nranks, port, m, n, r, iters, [None, None], q, "hals", 2, (3, 5, 2), False))
             for k in range(nranks)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(nranks))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][4], res[1][4])                      # V
    assert res[0][6] == res[1][6] and res[0][5] == res[1][5]         # sweep counts, costs
    costs = res[0][5]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(costs, costs[1:]))
