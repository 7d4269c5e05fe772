"""X H^T (nnf_xht_f32) per kernel form -- X fragments straight into registers (k_stream.hip, NNF_XHT=direct) against X staged
through LDS in 256-byte row pieces (k_xht_lds.hip, the default where it applies) -- at config B's shape and at the shape of
config D's partial product (T x_2 F2^T on the (I J) x K view), each checked against a float64 product on a slice.
    python tools/probes/xht_probe.py [m n r ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from nn_fac_amd.engine import get_engine
    eng = get_engine("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(1)
    for spec in sys.argv[2:]:
        m, n, r = (int(x) for x in spec.split("x"))
        X = torch.rand(m, n, device="cuda", generator=g)
        V = torch.rand(r, n, device="cuda", generator=g)
        out = eng.xht(X, V)
        sl = slice(max(0, m - 1000), m)
        ref = (V.double() @ X[sl].double().t())
        e = float((out[:, sl].double() - ref).norm() / ref.norm())
        e0 = float((out[:, :1000].double() - V.double() @ X[:1000].double().t()).norm() / ref.norm())
        for _ in range(3): eng.xht(X, V)
        ms = min(float(eng.time_kernel("xht", lambda: eng.xht(X, V))) for _ in range(3))
        by = (m * n + r * n + r * m) * 4.0
        print(f"  {os.environ.get('NNF_XHT', 'lds'):7s} {m:7d} x {n:5d} r={r:3d}: {1e3 * float(ms):7.1f} us  {by / float(ms) / 1e6:7.1f} GB/s  "
              f"{2.0 * m * n * r / float(ms) / 1e9:6.1f} TFLOP/s   rel err {e:.1e} / {e0:.1e}", flush=True)
    sys.exit(0)
specs = sys.argv[1:] or ["100000x2000x50", "250000x500x30", "100000x2000x64", "100000x2000x32", "100000x2000x16", "40000x5000x50",
                         "100001x1999x50", "777x130x20"]
for form in ("direct", None):
    env = dict(os.environ)
    env.pop("NNF_XHT", None)
    if form:
        env["NNF_XHT"] = form
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + specs, env=env, check=True)
