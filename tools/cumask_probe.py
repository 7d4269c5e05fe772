"""Do not run under rocprofv3: the profiler segfaults in __cxa_finalize on a process that owns CU-masked streams
(gpurun_out/prof_mask.log, round 1; the stream-creation interception of rocprofiler-sdk does not expect hipExtStreamCreateWithCUMask
streams at teardown).  Experiment dropped.

CU-masked streams (hipExtStreamCreateWithCUMask): does keeping the V-side sweep kernel and the cost kernel on disjoint
CUs remove the slowdown measured by tools/contention_probe.py?  Prints the solve time alone / next to the cost kernel for a
few mask layouts."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_fac_amd.engine import Engine
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
def masked_stream(bits):
    """bits: list of 0/1 per CU index."""
    words = (len(bits) + 31) // 32
    arr = (C.c_uint32 * words)()
    for i, b in enumerate(bits):
        if b: arr[i // 32] |= (1 << (i % 32))
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, arr)
    if rc != 0: raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value)
torch.cuda.set_device(0)
ncu = torch.cuda.get_device_properties(0).multi_processor_count
print("CUs", ncu, flush=True)
g = torch.Generator(device="cuda").manual_seed(0)
m, n, r = 100000, 2000, 50
X = torch.rand(m, n, device="cuda", generator=g); Ut = torch.rand(r, m, device="cuda", generator=g); V = torch.rand(r, n, device="cuda", generator=g)
W = torch.rand(400, r, device="cuda", generator=g); G = (W.t() @ W).contiguous(); M = torch.rand(r, n, device="cuda", generator=g) * 100
out = torch.empty(1, dtype=torch.float64, device="cuda")
engA, engB = Engine("cuda:0", workspace_bytes=64 << 20), Engine("cuda:0", workspace_bytes=64 << 20)
def run(sa, sb, tag):
    ts, tc = [], []
    for rep in range(4):
        Vc = V.clone(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if sb is not None:
            with torch.cuda.stream(sb):
                c0.record(sb)
                for _ in range(2): engB.frob_resid(X, Ut, V, out=out)
                c1.record(sb)
        with torch.cuda.stream(sa):
            a.record(sa); engA.hals_solve(M, G, Vc, 100, delta=0.0); b.record(sa)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
        if sb is not None: tc.append(c0.elapsed_time(c1) * 1e3 / 2)
    print(f"{tag:44s}: solve {min(ts[1:]):7.1f} us" + (f"   cost {min(tc[1:]):6.1f} us each" if tc else ""), flush=True)
plain_a, plain_b = torch.cuda.Stream(), torch.cuda.Stream()
run(plain_a, None, "alone, unmasked")
run(plain_a, plain_b, "next to cost, unmasked")
for name, sel in (("first 32 CUs | rest", lambda i: i < 32), ("every 8th CU | rest", lambda i: i % 8 == 0),
                  ("first 64 CUs | rest", lambda i: i < 64), ("every 4th CU | rest", lambda i: i % 4 == 0)):
    try:
        sa = masked_stream([1 if sel(i) else 0 for i in range(ncu)])
        sb = masked_stream([0 if sel(i) else 1 for i in range(ncu)])
        run(sa, None, f"alone, {name}")
        run(sa, sb, f"next to cost, {name}")
    except Exception as e:
        print(name, "failed:", e, flush=True)
