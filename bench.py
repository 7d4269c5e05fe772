#!/usr/bin/env python3
"""bench.py -- HALS NMF outer iterations/s on the MI355X engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--rule hals|mu] [--no-cpu] [--no-fixed]

One "step" = one outer NMF iteration (nn_fac.nmf.one_nmf_step semantics: U update, V update, cost) on synthetic dense
data already resident in HBM.  N = 1: configs[1] of BASELINE.json, 100000 x 2000 rank 50, HALS, fp32, deterministic
(alpha = inf, delta = 0.01, maxiter = 100).  N > 1 (torchrun, one rank per GPU, RCCL): the data matrix is row-sharded,
every rank holds a 100000 x 2000 block of an (N*100000) x 2000 matrix (weak scaling); the r x r Gram and r x n cross term
are all-reduced.  `value` counts 100000-row blocks processed per second = N * iterations/s, so it is
the whole-job aggregate and equals iterations/s at N = 1.

The JSON line also carries
  roofline     -- the dominant kernel (nnf_xty_kernel, "W^T X"): algorithmic flops 2*r*m*n per launch / its mean launch
                  duration, measured live with HIP events recorded on the launch stream right around the kernel (probe hook
                  of the C ABI), against the dense fp32 MFMA peak (157.3 TFLOP/s);
  cpu_baseline -- the NumPy restatement of the reference (oracle/, kind "port") timed on this box's host cores on a
                  bounded sample of the same workload -- 4 iterations of the full-size problem -- (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M, N, R = 100000, 2000, 50
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0


def synth_on_device(m, n, r, seed, device):
    """X = W*H* + 1e-2*rand (strictly positive), U0, V0 -- same recipe as SURVEY.md 8d, generated in HBM."""
    g = torch.Generator(device=device).manual_seed(seed)
    W = torch.rand(m, r, device=device, generator=g)
    H = torch.rand(r, n, device=device, generator=g)
    X = W @ H
    X += 1e-2 * torch.rand(m, n, device=device, generator=g)
    U0 = torch.rand(m, r, device=device, generator=g)
    V0 = torch.rand(r, n, device=device, generator=g)
    return X, U0, V0


def time_kernel(fn, reps, stream):
    """Mean duration (ms) of `fn` launches, HIP events recorded on the launch stream."""
    for _ in range(2):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    stream.synchronize()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def cpu_baseline(rule, beta):
    """Bounded sample of the same workload through the CPU oracle (the reference's statement sequence)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nnfac_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        threads = os.cpu_count() or 1
    ms = M                                       # the full 100000-row workload: ~10-20 s of CPU work on the GPU box's host
    X, U0, V0 = orc.synth_nmf(ms, N, R, seed=0, dtype=np.float32)
    U, V = U0, V0
    U, V, _ = orc.one_nmf_step(X, R, U, V, None, rule, beta, [None, None], [], [False, False], True)   # warm-up
    t0 = time.time()
    its = 4
    for _ in range(its):
        U, V, _ = orc.one_nmf_step(X, R, U, V, None, rule, beta, [None, None], [], [False, False], True)
    dt = (time.time() - t0) / its
    return {"value": 1.0 / dt, "unit": "iterations/s", "cores": int(threads), "kind": "port",
            "sample": f"{its} iterations (after 1 warm-up) of one_nmf_step on the full {ms}x{N} rank-{R} fp32 problem, "
                      f"same synthetic recipe; NumPy/OpenBLAS threads={threads}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rule", default="hals", choices=["hals", "mu"])
    ap.add_argument("--beta", type=float, default=None)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-fixed", action="store_true", help="skip the fixed-work line (profiling runs)")
    args = ap.parse_args()
    beta = args.beta if args.beta is not None else (2 if args.rule == "hals" else 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # NNF_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices)
        backend = os.environ.get("NNF_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(0)
    device = torch.device(f"cuda:{torch.cuda.current_device()}")

    from nn_fac_amd.engine import get_engine
    from nn_fac_amd import nmf as nmf_mod
    eng = get_engine(device)
    X, U0, V0 = synth_on_device(M, N, R, seed=rank, device=device)
    if world > 1:
        dist.broadcast(V0, src=0)               # V is replicated
    Ut, V = U0.t().contiguous(), V0.clone()
    group = dist.group.WORLD if world > 1 else None
    ws = nmf_mod._StepBuffers(X, R)
    sweeps = []

    def run(k):
        """k outer iterations through the product's own loop (nn_fac_amd.nmf.run_steps, the `for iteration` loop of
        compute_nmf: the cost + status block of every iteration is read back and handed to the stopping test's hook,
        here a recorder that never stops)."""
        nonlocal Ut, V
        last = [None]

        def retired(it, cost, sw):
            sweeps.append(sw)
            last[0] = cost
            return False

        Ut, V = nmf_mod.run_steps(eng, ws, X, R, Ut, V, k, args.rule, beta, [None, None], [], [False, False], True,
                                  retired, group=group)
        return last[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    sweeps.clear()
    barrier()
    t0 = time.perf_counter()
    cost = run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)

    sweeps_last = sweeps[-1] if sweeps else None
    sweeps_mean = float(np.mean([sum(x) for x in sweeps])) if sweeps else None
    final_cost = cost

    # fixed-work variant: 10 sweeps per inner solve whatever the data (delta = 0, maxiter = 10)
    fixed = None
    if args.rule == "hals" and not args.no_fixed:
        keep = dict(nmf_mod.HALS_INNER)
        nmf_mod.HALS_INNER.update(maxiter=10, delta=0.0)
        try:
            run(2)
            barrier()
            t1 = time.perf_counter()
            run(10)
            barrier()
            fdt = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([fdt], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                fdt = float(t)
            fixed = {"ms_per_step": 1e3 * fdt / 10, "iterations_per_s": world * 10 / fdt,
                     "inner": "10 sweeps per solve (delta=0, maxiter=10)"}
        finally:
            nmf_mod.HALS_INNER.update(keep)

    # dominant kernel, timed live on the launch stream: HIP events recorded by the library immediately around
    # nnf_xty_kernel (the probe hook of the C ABI), and around the whole nnf_xty_f32 call (kernel + slab reduction)
    stream = torch.cuda.current_stream(device)
    xty_call_ms = time_kernel(lambda: eng.xty(X, Ut, out=ws.UtM), 20, stream)
    reps = 20
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:          # torch creates the hipEvent_t lazily: record once so that the handles exist
        a.record(stream)
        b.record(stream)
    stream.synchronize()
    for a, b in evs:
        eng.set_probe(a, b)
        eng.xty(X, Ut, out=ws.UtM)
    eng.set_probe()
    stream.synchronize()
    xty_ms = sum(a.elapsed_time(b) for a, b in evs) / reps
    flops = 2.0 * R * M * N
    achieved = flops / (xty_ms * 1e-3) / 1e12
    xty_bytes = (M * N + R * M + R * N) * 4
    traffic = None      # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE), if present
    try:
        with open(os.path.join(ROOT, "profiles", "r01_xty_traffic.json")) as fh:
            traffic = json.load(fh)["hbm_bytes_per_launch"]
    except Exception:
        pass

    if rank == 0:
        out = {
            "metric": "HALS NMF outer iterations/s (100000x2000 rank-50 row blocks per second)" if args.rule == "hals"
                      else f"MU(beta={beta:g}) NMF outer iterations/s (100000x2000 rank-50 row blocks per second)",
            "value": world * args.steps / dt,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"NMF {args.rule} beta={beta:g}, {M}x{N} rank {R} per GPU "
                                   f"(configs[1] of BASELINE.json), deterministic (alpha=inf, delta=0.01, maxiter=100)",
                       "rows_total": world * M, "cols": N, "rank": R,
                       "parallelism": f"row-sharded x{world}" if world > 1 else "single GPU",
                       "inner_sweeps_per_step_last": sweeps_last,
                       "inner_sweeps_mean": sweeps_mean,
                       "final_cost": final_cost},
            "roofline": {"kernel": "nnf_xty_kernel (W^T X, the main kernel of nnf_xty_f32; its fixed-order slab reduction "
                                   "nnf_reduce_slabs_kernel follows and is included in call_ms)", "bound": "mfma",
                         "achieved": achieved, "call_ms": xty_call_ms,
                         "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_F32_PEAK_TFLOPS,
                         "traffic": traffic, "launch_ms": xty_ms,
                         "algorithmic_bytes": xty_bytes, "hbm_gbs": xty_bytes / (xty_ms * 1e-3) / 1e9,
                         "hbm_frac_of_8TBs": xty_bytes / (xty_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if fixed is not None:
            out["fixed_work"] = fixed
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.rule, beta)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
