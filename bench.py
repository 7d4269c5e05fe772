#!/usr/bin/env python3
"""bench.py -- outer iterations/s of the MI355X inner-update engine on BASELINE.json's configurations.

    python bench.py --gpus N --steps K --warmup W [--config B|C|D|E] [--no-cpu] [--no-fixed] [--no-extra]

One "step" = one outer iteration (nn_fac one_nmf_step / one_ntf_step semantics: every factor update + the cost) on
synthetic dense data already resident in HBM when the timed region starts.

  B (default)  100000 x 2000 rank 50 NMF, HALS, fp32, deterministic (alpha = inf, delta = 0.01, maxiter = 100) -- configs[1],
               the configuration BASELINE.json's metric is quoted on.  N > 1: X row-sharded, one 100000 x 2000 block per
               rank (weak scaling), RCCL all-reduce of the r x r Gram and r x n cross term;
               value = N * iterations/s = 100000-row blocks per second.
  C            same data, MU beta = 1 (configs[2]).
  D            500^3 rank-30 NTF, HALS, alpha = inf (configs[3]).  N > 1: leading mode sharded, one 500^3 block per rank.
  E            1e6 x 4000 rank 100 NMF, HALS (configs[4]): the SAME 1e6-row problem split over the N ranks (strong scaling;
               N = 1 holds all 16 GB).  Generated on the device in 8 row blocks of 125000 with per-block seeds, so every
               N factorises identical data.  The default (B) line carries a short E measurement under "extra_configs".

--gpus N > 1 without a torchrun environment: this process starts N ranks itself (python -m torch.distributed.run, one per
GPU, RCCL) BEFORE it touches the GPU, relays rank 0's JSON line and exits with the child's code.  Under torchrun
(WORLD_SIZE set) it is one of the ranks.  NNF_BENCH_BACKEND=gloo runs the ranks over gloo (ranks may then share devices:
a rehearsal, not a measurement).

The JSON line carries
  roofline      -- the dominant streaming kernel of the configuration (B/E: nnf_xty_kernel "W^T X"; C: nnf_mu_left_kernel;
                   D: the MTTKRP kernels): algorithmic flops or bytes per launch / mean launch duration, measured live
                   with HIP events recorded by the library on the launch stream right around that kernel;
  roofline_more -- the same for the other kernels of the step (the HALS sweep kernels against the fp32 VALU peak, X H^T,
                   the cost kernels, ...);
  cpu_baseline  -- the NumPy restatement of the reference (oracle/, kind "port") timed on this box's host cores on a
                   bounded sample of the same workload: the SAME X and the factors the GPU's timed region started from
                   (rank 0, N = 1 only).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md, "Peak FP32 (matrix)" = the fp32 vector peak
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    "B": dict(kind="nmf", rule="hals", beta=2, m=100000, n=2000, r=50, scaling="weak", ref="configs[1]"),
    "C": dict(kind="nmf", rule="mu", beta=1, m=100000, n=2000, r=50, scaling="weak", ref="configs[2]"),
    "D": dict(kind="ntf", rule="hals", beta=2, m=500, n=500, r=30, scaling="weak", ref="configs[3]"),
    "E": dict(kind="nmf", rule="hals", beta=2, m=1000000, n=4000, r=100, scaling="strong", ref="configs[4]"),
}
E_BLOCKS = 8            # config E is generated in 8 row blocks of m/8 rows (per-shard seeds)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS))
    ap.add_argument("--rule", default=None, choices=["hals", "mu"], help="alias: --rule mu = --config C")
    ap.add_argument("--beta", type=float, default=None)
    ap.add_argument("--shape", default=None, help="m,n,r override (tests / rehearsals; the line then says so)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-fixed", action="store_true", help="skip the fixed-work line (profiling runs)")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-E leg of the default line")
    ap.add_argument("--no-kernels", action="store_true", help="skip the per-kernel roofline probes")
    args = ap.parse_args(argv)
    if args.config is None:
        args.config = "C" if args.rule == "mu" else "B"
    return args


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """--gpus N > 1 outside torchrun: start N fresh ranks (nothing in this process has touched the GPU), relay their
    output and return the launcher's exit code.  Never an exec: this process stays the parent."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# ---- synthetic inputs --------------------------------------------------------------------------------
def synth_nmf_host(m, n, r, seed):
    """X = W*H* + 1e-2*rand (strictly positive), U0, V0 from one legacy RandomState(seed) stream, fp32 -- the recipe the CPU
    baseline leg reads too (same arrays, not a re-generation)."""
    import numpy as np
    rng = np.random.RandomState(seed)
    W, H = rng.rand(m, r), rng.rand(r, n)
    X = (W @ H + 1e-2 * rng.rand(m, n)).astype(np.float32)
    return X, rng.rand(m, r).astype(np.float32), rng.rand(r, n).astype(np.float32)


def synth_ntf_host(I, R, seed):
    import numpy as np
    rng = np.random.RandomState(seed)
    A, B, C = (rng.rand(I, R) for _ in range(3))
    kr = (A[:, None, :] * B[None, :, :]).reshape(-1, R)
    T = (kr @ C.T).reshape(I, I, I)
    T += 1e-2 * rng.rand(I, I, I)
    return T.astype(np.float32), [rng.rand(I, R).astype(np.float32) for _ in range(3)]


def synth_nmf_block_device(rows, n, r, seed, hseed, device, torch):
    """One row block of a device-generated problem (config E: its 16 GB never exist on the host): rows of W*, the noise and
    U0 from the block's own seed, H* from `hseed` (the same for every block of a problem)."""
    g = torch.Generator(device=device).manual_seed(seed)
    W = torch.rand(rows, r, device=device, generator=g)
    H = torch.rand(r, n, device=device, generator=torch.Generator(device=device).manual_seed(hseed))
    X = W @ H
    X += 1e-2 * torch.rand(rows, n, device=device, generator=g)
    U0 = torch.rand(rows, r, device=device, generator=g)
    return X, U0


# ---- CPU baseline ---------------------------------------------------------------------------------------------------
def cpu_baseline_nmf(X, U, V, r, rule, beta, gpu_sweeps, its=3, extras=True):
    """Bounded sample of the same workload through the CPU oracle (the reference's statement sequence): the same X, starting
    from the factors the GPU's timed region started from.  fp32 on all host threads (the headline), fp64 (the reference's
    default dtype) and a 1-thread figure on a row sample."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nnfac_oracle as orc
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        threadpool_limits, threads = None, os.cpu_count() or 1
    m, n = X.shape

    def run(Xd, Ud, Vd, its):
        sw = []
        t0 = time.time()
        for _ in range(its):
            log = []
            Ud, Vd, _ = orc.one_nmf_step(Xd, r, Ud, Vd, None, rule, beta, [None, None], [], [False, False], True,
                                         sweeps=log) if rule == "hals" else \
                orc.one_nmf_step(Xd, r, Ud, Vd, None, rule, beta, [None, None], [], [False, False], True)
            sw.append(log)
        return (time.time() - t0) / its, sw

    its = int(its)
    dt32, sw32 = run(X, U, V, its)
    out = {"value": 1.0 / dt32, "unit": "iterations/s", "cores": int(threads), "kind": "port",
           "sample": f"{its} iterations of one_nmf_step ({rule}, beta={beta:g}) on the full {m}x{n} rank-{r} fp32 problem: "
                     f"the same X as the GPU leg, starting from the factors its timed region started from; "
                     f"NumPy/OpenBLAS threads={threads}",
           "inner_sweeps": sw32, "gpu_inner_sweeps_same_iterations": gpu_sweeps[:its]}
    if not extras:
        return out
    dt64, _ = run(X.astype(np.float64), U.astype(np.float64), V.astype(np.float64), 1)
    out["fp64"] = {"value": 1.0 / dt64, "sample": "1 iteration, same start, float64 (the reference's default dtype)"}
    if threadpool_limits is not None:
        ms = max(1000, m // 16)
        with threadpool_limits(limits=1):
            dt1, _ = run(X[:ms], U[:ms], V, 1)
        out["one_thread"] = {"value": 1.0 / (dt1 * m / ms), "cores": 1,
                             "sample": f"1 iteration on the first {ms} rows (1/{m // ms} of the workload), time scaled by "
                                       f"{m / ms:.1f}"}
    return out


def cpu_baseline_nmf_shard(Xs, Us, V, r, rows_total, gpu_sweeps):
    """Config E's CPU leg: ONE iteration of the oracle's one_nmf_step on a row shard of the problem -- the first
    `Xs.shape[0]` rows, same data and start factors as the GPU leg -- scaled to the whole problem: everything that runs per row
    block (cross products, U-side solve, cost) scales with rows_total / rows; the replicated r x n V-side solve is timed on its
    own and counted once.  Labelled as a scaled shard figure, not a run of the whole problem."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nnfac_oracle as orc
    import math
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        threads = os.cpu_count() or 1
    ms = int(Xs.shape[0])
    log = []
    t0 = time.time()
    Un, Vn, _ = orc.one_nmf_step(Xs, r, Us, V, None, "hals", 2, [None, None], [], [False, False], True, sweeps=log)
    t_all = time.time() - t0
    UtU, UtM = Un.T @ Un, Un.T @ Xs
    t0 = time.time()
    orc.hals_nnls_acc(UtM, UtU, V, maxiter=100, alpha=math.inf, delta=0.01)
    t_v = time.time() - t0
    scale = rows_total / ms
    t_whole = scale * max(t_all - t_v, 0.0) + t_v
    return {"value": 1.0 / t_whole, "unit": "iterations/s", "cores": int(threads), "kind": "port",
            "sample": f"1 iteration of one_nmf_step (hals) on the first {ms} rows (1/{rows_total // ms} of the {rows_total} x "
                      f"{Xs.shape[1]} rank-{r} fp32 problem; same data and start factors as the GPU leg): {t_all:.2f} s, of which "
                      f"the replicated V-side solve {t_v:.2f} s; whole problem = {scale:.0f} x the per-row-block part + the V-side "
                      f"solve once = {t_whole:.1f} s per iteration -- a SCALED SHARD figure; NumPy/OpenBLAS threads={threads}",
            "inner_sweeps": log, "gpu_inner_sweeps_first_iteration": gpu_sweeps[:1]}


def cpu_baseline_ntf(T, F, R, its=2):
    import numpy as np
    import math
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nnfac_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        threads = os.cpu_count() or 1
    unf = [orc.unfold(T, k) for k in range(3)]
    nrm = float(np.linalg.norm(T.astype(np.float64)))
    its = int(its)
    t0 = time.time()
    for _ in range(its):
        F, _ = orc.one_ntf_step(unf, R, F, nrm, "hals", 2, [None] * 3, [], [False] * 3, alpha=math.inf)
    dt = (time.time() - t0) / its
    return {"value": 1.0 / dt, "unit": "iterations/s", "cores": int(threads), "kind": "port",
            "sample": f"{its} iterations of one_ntf_step (hals, alpha=inf) on the full {T.shape} rank-{R} fp32 tensor: the "
                      f"same tensor as the GPU leg, starting from the factors its timed region started from; "
                      f"NumPy/OpenBLAS threads={threads}"}


# ---- the ranks -------------------------------------------------------------------------------------------------------
class Ctx:
    """World / device plumbing of one rank."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = None
        factory = os.environ.get("NNF_BENCH_ENGINE")      # tests: "module:attr" of an engine double (CPU tensors)
        self.cuda = factory is None
        # NNF_BENCH_INIT_PG=1: initialise the process group even for a single rank (1-rank RCCL smoke on a one-GPU box)
        self.pg = self.world > 1 or (os.environ.get("NNF_BENCH_INIT_PG") == "1" and "MASTER_ADDR" in os.environ)
        if self.pg:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            self.backend = os.environ.get("NNF_BENCH_BACKEND", "nccl")
            if self.backend == "nccl":
                ndev = torch.cuda.device_count()
                if self.local_rank >= ndev:
                    raise SystemExit(f"bench.py: rank {self.rank} has no GPU of its own ({ndev} visible); RCCL needs one "
                                     f"device per rank (NNF_BENCH_BACKEND=gloo rehearses on shared devices)")
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{self.local_rank}"))
            else:
                if self.cuda:
                    torch.cuda.set_device(self.local_rank % max(1, torch.cuda.device_count()))
                dist.init_process_group(backend=self.backend)
        elif self.cuda:
            torch.cuda.set_device(0)
        self.device = torch.device(f"cuda:{torch.cuda.current_device()}") if self.cuda else torch.device("cpu")
        # NNF_BENCH_FORCE_SHARDED=1 (with NNF_BENCH_INIT_PG=1): the row-sharded protocol on ONE rank -- its chunked solves,
        # all-reduces and host decisions without any peer: what the protocol itself costs a rank on a one-GPU box
        self.force_sharded = self.pg and os.environ.get("NNF_BENCH_FORCE_SHARDED") == "1"
        self.group = dist.group.WORLD if (self.world > 1 or self.force_sharded) else None
        if self.force_sharded:
            # (the product's own switch -- NNF_FORCE_SHARDED, read when nn_fac_amd.dist is imported -- set here, so that the one
            #  variable of this script is enough: without it a one-rank group runs the UNSHARDED step)
            from nn_fac_amd import dist as _nd
            _nd.FORCE_SHARDED = True
        if factory is None:
            from nn_fac_amd.engine import get_engine
            self.eng = get_engine(self.device)
            self.dtype = torch.float32
        else:
            mod, attr = factory.split(":")
            self.eng = getattr(importlib.import_module(mod), attr)()
            self.dtype = torch.float64

    def barrier(self):
        if self.pg:
            self.dist.barrier()
        if self.cuda:
            self.torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        if self.pg:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    def spin_up(self, seconds=0.3):
        """Untimed, before the W warm-up steps: keep the device busy for `seconds` with a plain streaming op (a 256 MB torch scale
        in place: it shows up under its own name in a profile, not under a kernel of the product), so that the timed region does
        not start on a GPU that was idle a moment ago (a fresh process on a fresh box measured config D -- 7 ms of timed work -- at
        2560-2680 iterations/s, the same command run again at 2900-2940).  Touches no loop state."""
        if not self.cuda or getattr(self, "spun", False):
            return
        self.spun = True                       # once per process: the legs that follow the first find the device busy already
        scratch = self.torch.ones(64 << 20, dtype=self.torch.float32, device=self.device)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                scratch.mul_(1.0)
            self.torch.cuda.synchronize()
        del scratch

    def timed(self, fn):
        self.barrier()
        t0 = time.perf_counter()
        out = fn()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0), out


class InLoop:
    """HIP events recorded by the library right around the launches of ONE main kernel (nnf_ctx_set_probe_ring) INSIDE a
    region of the product's own loop -- next to whatever shares the chip with that kernel there (side-stream Grams and
    copies, the cost kernel under the V-side solve).  `times()` afterwards: one duration (ms) per launch, in launch order."""

    def __init__(self, cx, eng, kernel, npairs):
        self.eng, self.kernel = eng, kernel
        self.on = bool(cx.cuda) and hasattr(eng, "set_probe_ring") and npairs > 0
        self.pairs = eng.probe_pairs(npairs, cx.torch.cuda.current_stream(cx.device)) if self.on else []
        self.used = 0

    def __enter__(self):
        if self.on:
            self.eng.set_probe_ring(self.pairs, self.kernel)
        return self

    def __exit__(self, *exc):
        if self.on:
            self.used = self.eng.probe_ring_count()
            self.eng.set_probe_ring(None, self.kernel)
        return False

    def times(self):
        return [a.elapsed_time(b) for a, b in self.pairs[:self.used]]


class NmfRun:
    """The product's own outer loop (nn_fac_amd.nmf.run_steps = the `for iteration` loop of compute_nmf) on one rank's
    row block; every iteration's cost + status block is read back and handed to a recorder that never stops."""

    def __init__(self, cx, X, Ut, V, r, rule, beta, deterministic=True):
        from nn_fac_amd import nmf as nmf_mod
        self.cx, self.mod = cx, nmf_mod
        self.X, self.Ut, self.V, self.r, self.rule, self.beta = X, Ut, V, r, rule, beta
        self.ws = nmf_mod._StepBuffers(X, r, dtype=cx.dtype)
        self.sweeps, self.cost = [], None
        self.deterministic = deterministic

    def run(self, k):
        def retired(it, cost, sw):
            self.sweeps.append(sw)
            self.cost = cost
            return False
        self.Ut, self.V = self.mod.run_steps(self.cx.eng, self.ws, self.X, self.r, self.Ut, self.V, k, self.rule, self.beta,
                                             [None, None], [], [False, False], self.deterministic, retired,
                                             group=self.cx.group)
        return self.cost

    def measure(self, warmup, steps, probe=None):
        """`probe` = (kernel name, launches per step): that kernel's launches of the TIMED region are bracketed by events
        (InLoop); returns their durations as the fourth value."""
        self.cx.spin_up()
        self.run(warmup)
        self.sweeps.clear()
        start = (self.Ut.clone(), self.V.clone())
        with InLoop(self.cx, self.cx.eng, probe[0] if probe else "xty", steps * probe[1] if probe else 0) as il:
            dt, cost = self.cx.timed(lambda: self.run(steps))
        return dt, cost, start, il.times()

    def inloop(self, eng, kernel, per_step, steps):
        """`steps` more iterations of the same loop (not timed) with `kernel`'s launches on `eng` bracketed by events."""
        n0 = len(self.sweeps)
        with InLoop(self.cx, eng, kernel, steps * per_step + 2) as il:
            self.run(steps)
            if self.cx.cuda:
                self.cx.torch.cuda.synchronize()
        sw = self.sweeps[n0:]
        del self.sweeps[n0:]
        return il.times(), sw

    def fixed_work(self):
        """10 sweeps per inner solve whatever the data (delta = 0, maxiter = 10)."""
        keep = dict(self.mod.HALS_INNER)
        self.mod.HALS_INNER.update(maxiter=10, delta=0.0)
        try:
            self.run(2)
            fdt, _ = self.cx.timed(lambda: self.run(10))
        finally:
            self.mod.HALS_INNER.update(keep)
        return {"ms_per_step": 1e3 * fdt / 10, "iterations_per_s": 10 / fdt,
                "inner": "10 sweeps per solve (delta=0, maxiter=10)"}


PROFILE_TAG = "r04"      # profiles/<tag>_<config>_traffic.json: the committed PMC collection attach_traffic() may quote


def roof(kernel, bound, algo, ms, peak, unit, **more):
    """One roofline entry: `algo` algorithmic flops (bound mfma / valu) or bytes (bound hbm) per launch, `ms` mean launch."""
    achieved = algo / (ms * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
    d = {"kernel": kernel, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
         "traffic": None, "launch_ms": float(ms)}
    if hasattr(ms, "stats"):          # nothing is trimmed: mean over all samples + their spread
        d["launch_stats"] = ms.stats()
    d.update(more)
    return d


def library_build_flags():
    """The -D switches every translation unit of the loaded library was compiled with (nnf_build_flags)."""
    import ctypes
    from nn_fac_amd import _lib
    buf = ctypes.create_string_buffer(16384)
    n = _lib.load().nnf_build_flags(buf, 16384)
    return buf.value.decode() if 0 < n < 16384 else ""


def attach_traffic(entries, cfg_name, shape):
    """`traffic` of a roofline entry = HBM bytes per launch from the PMC passes of this same command committed under profiles/
    (tools/bench_profile.sh: separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` runs, FETCH_SIZE doubled per the gfx950
    correction of MI355X_MICROARCH.md, tools/collect_profiles.py) -- counters cannot be read from inside the process that is
    being timed.  The committed file names the shape and the library build switches it was collected with: the figure is
    attached only when both equal this run's (else `traffic` stays null and `traffic_note` says why); it is a figure of that
    collection, named as such in `traffic_source`, not a counter of this process."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{cfg_name}_traffic.json")
    if not os.path.exists(path):
        return
    try:
        with open(path) as fh:
            table = json.load(fh)
    except (OSError, ValueError):
        return
    meta = table.get("_meta") or {}
    why = None
    if meta.get("shape") != list(shape):
        why = f"committed traffic was collected at shape {meta.get('shape')}, this run is {list(shape)}"
    elif meta.get("build_flags") != library_build_flags():
        why = "committed traffic was collected with a library built with other -D switches"
    used = set()
    for e in entries:
        for key, row in table.items():
            if key.startswith("_") or key in used or not isinstance(row, dict):
                continue
            if e["kernel"].startswith(key) and row.get("hbm_bytes_per_launch"):
                if why:
                    e["traffic_note"] = why
                    break
                e["traffic"] = int(row["hbm_bytes_per_launch"])
                e["traffic_source"] = (f"profiles/{PROFILE_TAG}_{cfg_name}_traffic.json (PMC passes of this command at commit "
                                       f"{meta.get('commit', '?')}, mean of {row.get('launches')} launches)")
                if e.get("algorithmic_bytes"):
                    e["traffic_over_algorithmic"] = e["traffic"] / e["algorithmic_bytes"]
                used.add(key)
                break


def with_inloop(entry, times, algo, peak, unit, where):
    """Re-price a roofline entry on the launches of the product's loop: `launch_ms` / `achieved` / `frac` become the in-loop
    figures (mean over every launch recorded, nothing trimmed); the stand-alone figures move to `standalone`."""
    if not times:
        return entry
    from nn_fac_amd.engine import KernelTime
    kt = KernelTime.of(times)
    entry["standalone"] = {"launch_ms": entry["launch_ms"], "achieved": entry["achieved"], "frac": entry["frac"],
                           "launch_stats": entry.pop("launch_stats", None)}
    ach = algo / (float(kt) * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
    entry.update(launch_ms=float(kt), achieved=ach, frac=ach / peak, launch_stats=kt.stats(), measured=where)
    if "hbm_gbs" in entry and "algorithmic_bytes" in entry:
        entry["hbm_gbs"] = entry["algorithmic_bytes"] / float(kt) / 1e6
        if "hbm_frac_of_8TBs" in entry:
            entry["hbm_frac_of_8TBs"] = entry["hbm_gbs"] / HBM_PEAK_GBS
    return entry


def nmf_kernel_rooflines(cx, run, m, n, r, rule, beta, loop_times=None, steps=20):
    """Per-kernel launch durations, HIP events recorded by the library on the launch stream immediately around each main
    kernel.  Every entry is measured twice: stand-alone (Engine.time_kernel: back-to-back launches on the factors the timed
    region ended with) and on the launches INSIDE the product's loop (InLoop) -- `loop_times`: the dominant kernel's launches
    of the timed region itself; the other kernels: `steps` more iterations of the same loop per kernel, right after it.  The
    figures of an entry (`launch_ms`, `achieved`, `frac`) are the in-loop ones, the stand-alone ones sit under `standalone`."""
    eng, X, Ut, V, ws = cx.eng, run.X, run.Ut, run.V, run.ws
    where_timed = f"HIP events around this kernel's {len(loop_times or [])} launches inside the timed region"
    where_more = f"HIP events around this kernel's launches in {steps} further iterations of the same loop"
    torch = cx.torch
    flops = 2.0 * r * m * n
    xbytes = (m * n + r * m + r * n) * 4.0
    out = []
    if rule == "hals":
        ms = eng.time_kernel("xty", lambda: eng.xty(X, Ut, out=ws.UtM))
        out.append(roof("nnf_xty_kernel (W^T X, main kernel of nnf_xty_f32; the fixed-order slab reduction that follows is "
                        "not included)", "mfma", flops, ms, MFMA_F32_PEAK_TFLOPS, "TFLOP/s", algorithmic_flops=flops,
                        algorithmic_bytes=xbytes, hbm_gbs=xbytes / ms / 1e6, hbm_frac_of_8TBs=xbytes / ms / 1e6 / HBM_PEAK_GBS))
        with_inloop(out[-1], loop_times, flops, MFMA_F32_PEAK_TFLOPS, "TFLOP/s", where_timed)
        ms = eng.time_kernel("xht", lambda: eng.xht(X, V, out=ws.VMt))
        out.append(roof("nnf_xht_kernel (X H^T)", "mfma", flops, ms, MFMA_F32_PEAK_TFLOPS, "TFLOP/s",
                        algorithmic_flops=flops, algorithmic_bytes=xbytes, hbm_gbs=xbytes / ms / 1e6))
        with_inloop(out[-1], run.inloop(eng, "xht", 1, steps)[0], flops, MFMA_F32_PEAK_TFLOPS, "TFLOP/s", where_more)
        if ws.direct_cost:
            # the run has left the Gram-identity cost (flagged, or NNF_COST=direct): the streaming cost kernel of iteration i then
            # runs on the side stream BESIDE this kernel of iteration i+1 -- both launches stretch, their sum is what shrinks
            out[-1]["note"] = ("in these iterations the streaming cost kernel (the identity's error estimate exceeded its bound late "
                               "in the run) runs beside this kernel on the side stream: the in-loop duration is that of two "
                               "kernels sharing the chip; `standalone` is this kernel alone")
        c = torch.zeros(1, dtype=torch.float64, device=X.device)
        ms = eng.time_kernel("cost", lambda: eng.frob_resid(X, Ut, V, out=c))
        out.append(roof("nnf_cost_kernel<FROB> (||X - UV||^2 fused with the product)", "mfma", flops, ms,
                        MFMA_F32_PEAK_TFLOPS, "TFLOP/s", algorithmic_flops=flops, algorithmic_bytes=xbytes,
                        hbm_gbs=xbytes / ms / 1e6))
        # (the cost of iteration i runs beside the V-side solve of iteration i+1, on the cost stream's own context)
        ceng = ws.cost_eng if getattr(ws, "cost_eng", None) is not None else eng
        ctimes = run.inloop(ceng, "cost", 1, steps)[0]
        with_inloop(out[-1], ctimes, flops, MFMA_F32_PEAK_TFLOPS, "TFLOP/s", where_more + " (beside the V-side solve)")
        if not ctimes:
            out[-1]["measured"] = ("stand-alone only: the loop does not launch this kernel -- its cost comes from the Gram identity "
                                   "||X||^2 - 2<V,U^T X> + sum_j v_j^T (U^T U) v_j on the V update's operands (nnf_nmf_gram_cost_f32; "
                                   "this kernel is the fall-back for an almost exact fit, and NNF_COST=direct)")
        # the two persistent solves as the loop runs them (stopping rule on the device): launches alternate U side, V side
        ts, sw = run.inloop(eng, "hals", 2, steps)
        solves = {}
        cap = eng.hals_resident_columns(r) if hasattr(eng, "hals_resident_columns") else m
        if sw and len(ts) >= 2 * len(sw) and all(len(x) == 2 for x in sw) and m <= cap:   # (a solve = one launch)
            for side, name in ((0, "U"), (1, "V")):
                dur = [ts[2 * i + side] for i in range(len(sw))]
                cnt = [x[side] for x in sw]
                solves[name] = {"us_per_sweep": 1e3 * sum(dur) / max(1, sum(cnt)), "mean_launch_ms": sum(dur) / len(dur),
                                "mean_sweeps": sum(cnt) / len(cnt), "launches": len(dur)}
        # the sweep kernels: fixed 20 sweeps from the current factors (time / sweep is what the VALU roofline prices)
        ns = 20
        G = eng.gram(V)
        eng.xht(X, V, out=ws.VMt)
        # (more columns than the resident kernel holds -- config E on one device: the loop solves them in resident blocks, and
        # one such block is what this entry times)
        from nn_fac_amd import dist as _nd
        blk = _nd.column_blocks(eng, Ut)[0]
        cols_blk = blk[1] - blk[0]
        F = Ut[:, :cols_blk].clone() if cols_blk < m else Ut.clone()
        VMb = ws.VMt[:, :cols_blk] if cols_blk < m else ws.VMt
        ms = eng.time_kernel("hals", lambda: eng.hals_sweeps(VMb, G, F, ns), reps=5) / ns
        m_all, m = m, cols_blk
        on_mfma = hasattr(eng, "hals_resid_floats") and eng.hals_resid_floats(r, cols_blk) > 0
        if on_mfma:
            out.append(roof(f"nnf_hals_mfma_kernel (U side: {m} columns, rank {r}; push form of the sweep on the matrix cores, "
                            f"k_hals_mfma.hip, per sweep over {ns} fixed sweeps)", "mfma", 2.0 * r * r * m, ms, MFMA_F32_PEAK_TFLOPS,
                            "TFLOP/s", algorithmic_flops=2.0 * r * r * m, us_per_sweep=ms * 1e3, in_loop_solve=solves.get("U"),
                            note="2 r^2 flops per column and sweep as rank-4 MFMA updates of the scaled residual; the four row "
                                 "updates per k-block stay on the VALU, which never overlaps fp32 MFMAs on a SIMD (DESIGN.md 3); "
                                 "in_loop_solve: the persistent solve of the loop, stopping rule included"))
        else:
            out.append(roof(f"nnf_hals_kernel (U side: {m} columns, rank {r}; one lane per column, per sweep over {ns} fixed "
                            f"sweeps)", "valu", 2.0 * r * r * m, ms, MFMA_F32_PEAK_TFLOPS, "TFLOP/s",
                            algorithmic_flops=2.0 * r * r * m, us_per_sweep=ms * 1e3, in_loop_solve=solves.get("U"),
                            note="Gauss-Seidel row dependence: issue bound of this formulation is ~2.4x the 2r^2m/peak time "
                                 "(DESIGN.md 3); in_loop_solve: the persistent solve of the loop, stopping rule included"))
        m = m_all
        G2 = eng.gram(Ut)
        eng.xty(X, Ut, out=ws.UtM)
        F2 = V.clone()
        ms = eng.time_kernel("hals", lambda: eng.hals_sweeps(ws.UtM, G2, F2, ns), reps=5) / ns
        out.append(roof(f"V-side sweep kernel ({n} columns, rank {r}; few-column layout, per sweep over {ns} fixed sweeps)", "valu",
                        2.0 * r * r * n, ms, MFMA_F32_PEAK_TFLOPS, "TFLOP/s", algorithmic_flops=2.0 * r * r * n,
                        us_per_sweep=ms * 1e3, in_loop_solve=solves.get("V"),
                        note="fixed-count launches (nnf_hals_sweeps_f32: four lanes per column), latency bound; in_loop_solve: the "
                             "persistent solve of the loop -- nnf_hals_wave_kernel, one wave per column, 1.5-1.7 us per sweep"))
    else:
        k = 2.0 if float(beta) != 2.0 else 1.0
        c = torch.zeros(1, dtype=torch.float64, device=X.device)
        if float(beta) == 1.0 and r <= eng.MU_FUSED_MAX_RANK:
            # the form the timed loop runs: the left update that also returns the KL divergence of its input factors (the
            # previous iteration's cost).  Same 4 r m n MFMA flops; the divergence terms (a series + a logarithm per entry)
            # are VALU work on the same fp32 pipe and are NOT counted as algorithmic flops.
            ms = eng.time_kernel("mu_left", lambda: eng.mu_left(X, Ut, V, beta, cost_out=c))
            out.append(roof("nnf_mu_left_kernel<KL + cost> (beta=1: P = U V, num += (X ./ P) V^T and the KL divergence of the "
                            "input factors in one pass; the kernel of the timed loop)", "mfma", k * flops, ms,
                            MFMA_F32_PEAK_TFLOPS, "TFLOP/s", algorithmic_flops=k * flops, algorithmic_bytes=xbytes,
                            hbm_gbs=xbytes / ms / 1e6))
            # (the first launch of a run has no previous cost to form: the plain kernel; every later one is this form)
            with_inloop(out[-1], (loop_times or [])[1:], k * flops, MFMA_F32_PEAK_TFLOPS, "TFLOP/s", where_timed)
        ms = eng.time_kernel("mu_left", lambda: eng.mu_left(X, Ut, V, beta))
        out.append(roof(f"nnf_mu_left_kernel (beta={beta:g}: P = U V and num += (X ./ P) V^T in one pass)", "mfma",
                        k * flops, ms, MFMA_F32_PEAK_TFLOPS, "TFLOP/s", algorithmic_flops=k * flops,
                        algorithmic_bytes=xbytes, hbm_gbs=xbytes / ms / 1e6))
        ms = eng.time_kernel("mu_right", lambda: eng.mu_right(X, Ut, V, beta))
        out.append(roof(f"nnf_mu_right_kernel (beta={beta:g})", "mfma", k * flops, ms, MFMA_F32_PEAK_TFLOPS, "TFLOP/s",
                        algorithmic_flops=k * flops, algorithmic_bytes=xbytes, hbm_gbs=xbytes / ms / 1e6))
        with_inloop(out[-1], run.inloop(eng, "mu_right", 1, steps)[0], k * flops, MFMA_F32_PEAK_TFLOPS, "TFLOP/s", where_more)
        ms = eng.time_kernel("cost", lambda: eng.betadiv(X, Ut, V, beta, out=c))
        out.append(roof(f"nnf_cost_kernel<beta={beta:g}> (divergence fused with the product)", "mfma", flops, ms,
                        MFMA_F32_PEAK_TFLOPS, "TFLOP/s", algorithmic_flops=flops, algorithmic_bytes=xbytes,
                        hbm_gbs=xbytes / ms / 1e6))
    return out


def bench_nmf(cx, args, cfg, steps, warmup, with_cpu, with_fixed, with_kernels, cpu_its=3, steady=False):
    torch = cx.torch
    m, n, r, rule = cfg["m"], cfg["n"], cfg["r"], cfg["rule"]
    beta = args.beta if (args.beta is not None and rule == "mu") else cfg["beta"]
    host = None
    if cfg["scaling"] == "strong":
        # config E: blocks [b0, b1) of the E_BLOCKS row blocks live on this rank
        if E_BLOCKS % cx.world:
            raise SystemExit(f"config E is laid out in {E_BLOCKS} row blocks: --gpus must divide {E_BLOCKS}")
        per = E_BLOCKS // cx.world
        rows = m // E_BLOCKS
        parts = [synth_nmf_block_device(rows, n, r, args.seed * E_BLOCKS + b, 977 + args.seed, cx.device, torch)
                 for b in range(cx.rank * per, (cx.rank + 1) * per)]
        X = torch.cat([p[0] for p in parts]) if per > 1 else parts[0][0]
        Ut = (torch.cat([p[1] for p in parts]) if per > 1 else parts[0][1]).t().contiguous()
        del parts
        V = torch.rand(r, n, device=cx.device, generator=torch.Generator(device=cx.device).manual_seed(4242 + args.seed))
        units = 1.0                      # one iteration of the whole job per step, whatever N
    else:
        Xh, U0h, V0h = synth_nmf_host(m, n, r, seed=args.seed + cx.rank)
        X = torch.from_numpy(Xh).to(device=cx.device, dtype=cx.dtype)
        Ut = torch.from_numpy(U0h).to(device=cx.device, dtype=cx.dtype).t().contiguous()
        V = torch.from_numpy(V0h).to(device=cx.device, dtype=cx.dtype)
        if cx.world > 1:
            cx.dist.broadcast(V, src=0)          # V is replicated
        host = Xh if (with_cpu and cx.world == 1 and cx.rank == 0) else None
        del Xh, U0h, V0h
        units = float(cx.world)          # 100000-row blocks per step

    run = NmfRun(cx, X, Ut, V, r, rule, beta)
    probe = ("xty" if rule == "hals" else "mu_left", 1) if (with_kernels and cx.cuda and cx.rank == 0) else None
    dt, cost, start, loop_times = run.measure(warmup, steps, probe=probe)
    sweeps = list(run.sweeps)
    if cx.group is not None and os.environ.get("NNF_BENCH_DEBUG"):
        print(f"[rank {cx.rank}] sharded U-side protocol: {run.ws.async_hits} device-side decisions, "
              f"{run.ws.async_misses} redone synchronously; sweeps {sweeps}", file=sys.stderr, flush=True)
    out = {"value": units * steps / dt, "ms_per_step": 1e3 * dt / steps, "final_cost": cost,
           "inner_sweeps_per_step_last": sweeps[-1] if sweeps else None,
           "inner_sweeps_mean": (sum(sum(x) for x in sweeps) / len(sweeps)) if sweeps and sweeps[0] else None,
           "rows_per_rank": int(X.shape[0]), "rule": rule, "beta": beta}
    if cx.group is not None:
        from nn_fac_amd import dist as nd
        out["sharded_protocol"] = {
            "u_side_stopping_decision": "device (NNF_SHARDED_ASYNC=1)" if run.ws.async_sharded else "host-synchronous (default)",
            "cost": "overlapped with the next V-side solve (NNF_SHARDED_OVERLAP=1)"
                    if nd.opt_in("NNF_SHARDED_OVERLAP", cx.group) else "inside the step (default)",
            "device_decisions": run.ws.async_hits, "redone_synchronously": run.ws.async_misses,
            "fell_back_to_chunked_solves": bool(run.ws.safe_solve)}
    if rule == "hals" and cx.cuda:
        out["cost_evaluation"] = ("streaming kernel (nnf_frob_resid_f32)" if (run.ws.direct_cost or os.environ.get("NNF_COST") == "direct")
                                  else "Gram identity (nnf_nmf_gram_cost_f32), fp64 inner products, guarded by its own error estimate")
    if steady and rule == "hals" and cx.world == 1:
        # a window that starts once the inner solves have settled (two consecutive iterations whose sweep counts differ by <= 4
        # in both solves -- the loop's own settle rule): comparable across boxes and rounds, unlike the default window, which
        # sits in the transient of the first two dozen iterations
        extra_its, prev = 0, sweeps[-1] if sweeps else None
        while extra_its < 80:
            run.sweeps.clear()
            run.run(2)
            extra_its += 2
            a_, b_ = run.sweeps[-2], run.sweeps[-1]
            if all(abs(x - y) <= 4 for x, y in zip(a_, b_)):
                break
        run.sweeps.clear()
        sdt, _ = cx.timed(lambda: run.run(steps))
        ssw = list(run.sweeps)
        out["steady_state"] = {"iterations_per_s": units * steps / sdt, "ms_per_step": 1e3 * sdt / steps, "steps": steps,
                               "iterations_before_the_window": warmup + steps + extra_its,
                               "inner_sweeps_mean": sum(sum(x) for x in ssw) / max(1, len(ssw)),
                               "inner_sweeps_per_step_last": ssw[-1] if ssw else None,
                               "rule": "starts once two consecutive iterations differ by <= 4 sweeps in both inner solves"}
    if with_fixed and rule == "hals":
        f = run.fixed_work()
        f["iterations_per_s"] *= units
        out["fixed_work"] = f
    if with_kernels and cx.cuda and cx.rank == 0:
        out["rooflines"] = nmf_kernel_rooflines(cx, run, int(X.shape[0]), n, r, rule, beta, loop_times, steps)
    if with_fixed and rule == "hals" and cx.world == 1 and cfg["scaling"] == "weak":
        # the reference's DEFAULT call, nmf(..., deterministic=False) (nn_fac/nmf.py:19-22): the inner solves stop on the
        # wall-clock rule cnt <= 1 + 0.5 * rho (nnls.py:156,190-194) -- per HALS solve two device syncs around the Gram + cross
        # launches and a one-sweep probe on a scratch copy -- so the result is time dependent by design, as in the reference
        nd_run = NmfRun(cx, X, start[0].clone(), start[1].clone(), r, rule, beta, deterministic=False)
        nsteps = min(steps, 10)
        ndt, ncost, _, _ = nd_run.measure(2, nsteps)
        out["nondeterministic"] = {"what": "the same data through deterministic=False, the reference's default (wall-clock sweep budget)",
                                   "iterations_per_s": units * nsteps / ndt, "ms_per_step": 1e3 * ndt / nsteps,
                                   "steps": nsteps, "warmup": 2, "final_cost": ncost,
                                   "inner_sweeps_per_step_last": nd_run.sweeps[-1] if nd_run.sweeps else None}
        del nd_run
    if host is not None:
        U_s, V_s = start[0].t().contiguous().cpu().numpy(), start[1].cpu().numpy()
        del run, X
        out["cpu_baseline"] = cpu_baseline_nmf(host, U_s, V_s, r, rule, beta, sweeps, its=cpu_its, extras=cpu_its >= 3)
    elif with_cpu and cfg["scaling"] == "strong" and cx.world == 1 and cx.rank == 0 and cx.cuda:
        ms = m // 8                  # one of the 8 row blocks (what a rank of the 8-GPU run holds): ~10 s of CPU work
        out["cpu_baseline"] = cpu_baseline_nmf_shard(X[:ms].cpu().numpy(), start[0][:, :ms].t().contiguous().cpu().numpy(),
                                                     start[1].cpu().numpy(), r, m, sweeps)
    return out


def bench_ntf(cx, args, cfg, steps, warmup, with_cpu, with_kernels, cpu_its=2):
    import math
    torch = cx.torch
    from nn_fac_amd import ntf as ntf_mod
    I, R = cfg["m"], cfg["r"]
    Th, F0h = synth_ntf_host(I, R, seed=args.seed + cx.rank)
    T = torch.from_numpy(Th).to(device=cx.device, dtype=cx.dtype)
    Ft = [torch.from_numpy(f).to(device=cx.device, dtype=cx.dtype).t().contiguous() for f in F0h]
    if cx.world > 1:
        for f in Ft[1:]:
            cx.dist.broadcast(f, src=0)          # the factors of the unsharded modes are replicated
    st = ntf_mod._NtfState(cx.eng, T, group=cx.group)
    sweeps, last = [], [None]

    def retired(it, cost, sw):
        sweeps.append(sw)
        last[0] = cost
        return False

    def run(k):
        nonlocal Ft
        Ft = ntf_mod.run_ntf_steps(st, R, Ft, k, "hals", 2, [None] * 3, [], [False] * 3, math.inf, 0.01, retired)
        return last[0]

    cx.spin_up()
    run(warmup)
    sweeps.clear()
    start = [f.clone() for f in Ft]
    # the timed region carries no probe events: at 0.34 ms per iteration the two stream markers of a bracketed launch cost 12 % of the
    # rate (2561 against 2908 iterations/s, same box); the kernels' in-loop launches are bracketed in further passes of the same loop
    dt, cost = cx.timed(lambda: run(steps))
    out = {"value": cx.world * steps / dt, "ms_per_step": 1e3 * dt / steps, "final_cost": cost,
           "inner_sweeps_per_step_last": sweeps[-1] if sweeps else None,
           "inner_sweeps_mean": (sum(sum(x) for x in sweeps) / len(sweeps)) if sweeps else None,
           "rows_per_rank": I, "rule": "hals", "beta": 2,
           "cost_evaluation": ("pass over T (nnf_cp3_partial_cost_f32)" if (st.direct_cost or os.environ.get("NNF_COST") == "direct")
                               else "the reference's expression ||T||^2 - 2<F,rhs> + sum f^T cross f on the last mode's operands "
                                    "(nnf_nmf_gram_cost_f32), fp64 inner products, guarded by its own error estimate")}
    if with_kernels and cx.cuda and cx.rank == 0:
        eng = cx.eng
        tb = I * I * I * 4.0 + 3 * I * R * 4.0
        fl = 2.0 * I * I * I * R
        rl = []
        # the two passes over T of an iteration of the loop timed above (dimension tree + identity cost):
        ms = eng.time_kernel("mttkrp", lambda: eng.mttkrp3(T, Ft, 2))
        rl.append(roof("nnf_mttkrp_rows_kernel (mode-2 MTTKRP, Khatri-Rao operand generated on the fly; slab reduction not "
                       "included)", "hbm", tb, ms, HBM_PEAK_GBS, "GB/s", algorithmic_bytes=tb, algorithmic_flops=fl,
                       tflops=fl / ms / 1e9))
        with InLoop(cx, eng, "mttkrp", steps + 2) as il_m:
            run(steps)
            torch.cuda.synchronize()
        with_inloop(rl[-1], il_m.times()[-steps:], tb, HBM_PEAK_GBS, "GB/s",
                    f"HIP events around this kernel's launches in {steps} further iterations of the same loop")
        rl[-1]["tflops"] = fl / rl[-1]["launch_ms"] / 1e9
        # the partial product Y = T x_2 F2^T the mode-0 / mode-1 right-hand sides are contracted from: the X H^T kernel on the
        # (I J) x K unfolding -- a view of T.  (The cost comes from the last mode's operands: nnf_gram_cost_kernel, ~10 us.)
        yb = tb - 2 * I * R * 4.0 + R * I * I * 4.0
        with InLoop(cx, eng, "xht", steps + 2) as il_y:
            run(steps)
            torch.cuda.synchronize()
        Y = torch.empty((R, I, I), dtype=torch.float32, device=T.device)
        ms = eng.time_kernel("xht", lambda: eng.ttm3(T, Ft[2], 2, out=Y))
        rl.append(roof("nnf_xht_lds_kernel as T x_2 F2^T (the partial product Y, R x I x J, of the dimension tree)", "hbm", yb, ms,
                       HBM_PEAK_GBS, "GB/s", algorithmic_bytes=yb, algorithmic_flops=fl, tflops=fl / ms / 1e9))
        with_inloop(rl[-1], il_y.times()[-steps:], yb, HBM_PEAK_GBS, "GB/s",
                    f"HIP events around this kernel's launches in {steps} further iterations of the same loop")
        c = torch.zeros(3, dtype=torch.float64, device=T.device)
        ms = eng.time_kernel("mu_left", lambda: eng.cp3_partial_cost(T, Ft, Y, c[0:1]))
        rl.append(roof("nnf_mu_left_kernel<FROB> (one pass: ||T - model||^2 AND the partial product; what the loop runs instead "
                       "once the Gram-identity cost is flagged unreliable or the stopping test is near -- not in the timed loop)",
                       "mfma", 2 * fl, ms, MFMA_F32_PEAK_TFLOPS, "TFLOP/s", algorithmic_flops=2 * fl,
                       algorithmic_bytes=tb + R * I * I * 4.0, hbm_gbs=(tb + R * I * I * 4.0) / ms / 1e6))
        # the kernels the fused pass replaces / the first iteration and one_ntf_step use
        for mode in range(2):
            ms = eng.time_kernel("mttkrp", lambda: eng.mttkrp3(T, Ft, mode))
            rl.append(roof(f"nnf_mttkrp_seg_kernel, mode {mode} (direct MTTKRP; not in the timed loop since the dimension tree)",
                           "hbm", tb, ms, HBM_PEAK_GBS, "GB/s", algorithmic_bytes=tb, algorithmic_flops=fl,
                           tflops=fl / ms / 1e9))
        ms = eng.time_kernel("cost", lambda: eng.cp3_betadiv(T, Ft, 2, out=c[0:1]))
        rl.append(roof("nnf_cost_kernel<FROB> on the CP model (stand-alone cost; one_ntf_step)", "hbm", tb, ms, HBM_PEAK_GBS,
                       "GB/s", algorithmic_bytes=tb, algorithmic_flops=fl, tflops=fl / ms / 1e9))
        out["rooflines"] = rl
    if with_cpu and cx.world == 1 and cx.rank == 0:
        Fs = [f.t().contiguous().cpu().numpy() for f in start]
        del T, st
        out["cpu_baseline"] = cpu_baseline_ntf(Th, Fs, R, its=cpu_its)
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))          # before anything here has touched the GPU

    cx = Ctx(args)
    cfg = dict(CONFIGS[args.config])
    shape_note = ""
    if args.shape:
        cfg["m"], cfg["n"], cfg["r"] = (int(x) for x in args.shape.split(","))
        shape_note = " [--shape override: NOT the BASELINE configuration]"
    world = cx.world

    if cfg["kind"] == "nmf":
        res = bench_nmf(cx, args, cfg, args.steps, args.warmup, not args.no_cpu, not args.no_fixed, not args.no_kernels,
                        steady=(args.config == "B" and not args.no_fixed and not args.shape))
    else:
        res = bench_ntf(cx, args, cfg, args.steps, args.warmup, not args.no_cpu, not args.no_kernels)

    extra = None
    if args.config == "B" and not args.no_extra and not args.shape:
        # the 1e6 x 4000 rank-100 problem of configs[4] split over these N ranks (strong scaling), a short measurement
        ecfg = dict(CONFIGS["E"])
        if cx.cuda:
            cx.torch.cuda.empty_cache()
        ewhat = (f"NMF hals {ecfg['m']}x{ecfg['n']} rank {ecfg['r']} ({ecfg['ref']} of BASELINE.json), the same 1e6-row problem "
                 f"row-sharded over {world} rank(s): strong scaling")
        try:
            e = bench_nmf(cx, args, ecfg, 5, 2, False, False, False)
            extra = {"E": {"workload": ewhat, "iterations_per_s": e["value"], "ms_per_step": e["ms_per_step"], "steps": 5,
                           "warmup": 2, "rows_per_rank": e["rows_per_rank"],
                           "inner_sweeps_per_step_last": e["inner_sweeps_per_step_last"], "final_cost": e["final_cost"],
                           "scaling": "strong"}}
        except Exception as exc:      # an extra: it must never cost the line of the configuration that was asked for
            # (deterministic failures -- an unsupported shape, an allocation -- hit every rank at the same call)
            extra = {"E": {"workload": ewhat, "error": f"{type(exc).__name__}: {exc}"[:300]}}
        # what ONE rank of the 8-GPU run of configs[4] computes per iteration: a 125000 x 4000 rank-100 row block (the same
        # generator, block 0), here WITHOUT the sharded protocol's exchanges (a one-GPU box has no peer): the compute side of the
        # per-rank step, measured instead of estimated; profiles/r04_E_block_* holds the same block under the forced sharded
        # protocol on a 1-rank RCCL group with its kernel breakdown
        try:
            if world != 1:
                raise RuntimeError("measured on one rank only")
            if cx.cuda:
                cx.torch.cuda.empty_cache()
            bcfg = dict(CONFIGS["E"], m=CONFIGS["E"]["m"] // E_BLOCKS)
            b = bench_nmf(cx, args, bcfg, 10, 3, False, False, False)
            extra["E_rank_block_of_8"] = {
                "workload": f"NMF hals {bcfg['m']}x{bcfg['n']} rank {bcfg['r']}: one rank's row block of configs[4] at 8 GPUs, "
                            f"unsharded loop (no collectives), 10 steps after 3 warm-up",
                "iterations_per_s": b["value"], "ms_per_step": b["ms_per_step"], "steps": 10, "warmup": 3,
                "inner_sweeps_per_step_last": b["inner_sweeps_per_step_last"], "final_cost": b["final_cost"]}
        except Exception as exc:
            extra["E_rank_block_of_8"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        # configs[2] and configs[3], short legs on the same box in the same run: rate, the dominant kernel's roofline (launches
        # inside the leg's timed loop) and a one-iteration CPU baseline
        for name in ("C", "D"):
            xcfg = dict(CONFIGS[name])
            if cx.cuda:
                cx.torch.cuda.empty_cache()
            xwhat = (f"{'NMF mu beta=1' if name == 'C' else 'NTF hals'} {xcfg['m']}{'x' + str(xcfg['n']) if name == 'C' else '^3'} rank "
                     f"{xcfg['r']} per GPU ({xcfg['ref']} of BASELINE.json), 10 steps after 3 warm-up")
            try:
                with_x = world == 1 and not args.no_cpu
                if xcfg["kind"] == "nmf":
                    x = bench_nmf(cx, args, xcfg, 10, 3, with_x, False, not args.no_kernels, cpu_its=1)
                else:
                    x = bench_ntf(cx, args, xcfg, 10, 3, with_x, not args.no_kernels, cpu_its=1)
                rlx = x.get("rooflines") or []
                extra[name] = {"workload": xwhat, "iterations_per_s": x["value"], "ms_per_step": x["ms_per_step"], "steps": 10,
                               "warmup": 3, "inner_sweeps_per_step_last": x.get("inner_sweeps_per_step_last"),
                               "final_cost": x["final_cost"], "scaling": "weak", "roofline": rlx[0] if rlx else None,
                               "cpu_baseline": x.get("cpu_baseline")}
            except Exception as exc:
                extra[name] = {"workload": xwhat, "error": f"{type(exc).__name__}: {exc}"[:300]}

    if cx.rank == 0:
        rule, beta = res["rule"], res["beta"]
        what = {"B": "HALS NMF outer iterations/s (100000x2000 rank-50 row blocks per second)",
                "C": f"MU(beta={beta:g}) NMF outer iterations/s (100000x2000 rank-50 row blocks per second)",
                "D": "HALS NTF outer iterations/s (500^3 rank-30 tensor blocks per second)",
                "E": "HALS NMF outer iterations/s on the 1e6x4000 rank-100 problem (whole job)"}[args.config]
        if cfg["kind"] == "nmf":
            workload = (f"NMF {rule} beta={beta:g}, {cfg['m']}x{cfg['n']} rank {cfg['r']} "
                        + ("per GPU" if cfg["scaling"] == "weak" else f"in total, row-sharded over {world} GPU(s)")
                        + f" ({cfg['ref']} of BASELINE.json), deterministic (alpha=inf, delta=0.01, maxiter=100)")
        else:
            workload = (f"NTF hals, {cfg['m']}^3 rank {cfg['r']} per GPU ({cfg['ref']} of BASELINE.json), alpha=inf, "
                        f"delta=0.01, maxiter=100")
        rl = res.get("rooflines") or []
        if rl and not args.shape and world == 1:
            attach_traffic(rl, args.config, (cfg["m"], cfg["n"], cfg["r"]))
        out = {
            "metric": what + shape_note,
            "value": res["value"],
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": cfg["scaling"],
            "vs_baseline": None,
            "dtype": "f32" if cx.cuda else "f64 (CPU engine double: a launch rehearsal, not a measurement)",
            "data": "synthetic",
            "config": {"workload": workload + shape_note,
                       "rows_total": (world if cfg["scaling"] == "weak" else 1) * cfg["m"], "cols": cfg["n"],
                       "rank": cfg["r"], "rows_per_rank": res["rows_per_rank"],
                       "parallelism": (f"row-sharded x{world} ({cx.backend})" if world > 1 else
                                       "single GPU" + (f" (process group: {cx.backend}, 1 rank)" if cx.pg else "")),
                       "inner_sweeps_per_step_last": res["inner_sweeps_per_step_last"],
                       "inner_sweeps_mean": res["inner_sweeps_mean"],
                       "final_cost": res["final_cost"]},
            "roofline": rl[0] if rl else None,
            "roofline_more": rl[1:],
        }
        if "fixed_work" in res:
            out["fixed_work"] = res["fixed_work"]
        if "nondeterministic" in res:
            out["nondeterministic"] = res["nondeterministic"]
        if "steady_state" in res:
            out["steady_state"] = res["steady_state"]
        if "cost_evaluation" in res:
            out["config"]["cost_evaluation"] = res["cost_evaluation"]
        if getattr(cx, "spun", False):
            out["config"]["untimed_spin_up"] = ("0.3 s of a plain 256 MB in-place scale (torch) before the W warm-up steps: the timed region "
                                                "does not start on a GPU that was idle a moment ago; not part of W or K")
        if "sharded_protocol" in res:
            out["config"]["sharded_protocol"] = res["sharded_protocol"]
        if extra is not None:
            out["extra_configs"] = extra
        if "cpu_baseline" in res:
            out["cpu_baseline"] = res["cpu_baseline"]
        print(json.dumps(out), flush=True)
    if cx.pg:
        cx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
