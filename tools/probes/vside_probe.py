"""Few-column persistent HALS solves (the V side of NMF, NTF factors, the replicated solve of the sharded problem): us per sweep
of a 100-sweep solve (delta = 0: every sweep pays the grid exchange) per column layout -- four lanes per column
(k_hals_quad.hip, NNF_HALS_FORCE=quad) against one wave per column (k_hals_wave.hip, the default up to 8192 columns) --
alone and next to the Frobenius cost kernel of a 100000 x 2000 problem on a second stream (what the V-side solve of the NMF
loop runs beside).  One process per layout (the switch is read from the environment at every call, but keep it simple).
    python tools/probes/vside_probe.py [rxn ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from nn_fac_amd.engine import get_engine, get_side_engine
    eng = get_engine()
    seng, sstream = get_side_engine("cuda:0", "cost")
    g = torch.Generator(device="cuda").manual_seed(0)
    m, n0, r0 = 100000, 2000, 50
    X = torch.rand(m, n0, device="cuda", generator=g); Ut = torch.rand(r0, m, device="cuda", generator=g); V0 = torch.rand(r0, n0, device="cuda", generator=g)
    out = torch.empty(1, dtype=torch.float64, device="cuda")
    main = torch.cuda.current_stream()
    for spec in sys.argv[2:]:
        r, n = (int(x) for x in spec.split("x"))
        W = torch.rand(4 * r, r, device="cuda", generator=g); G = (W.t() @ W).contiguous()
        M = (W.t() @ (W @ torch.rand(r, n, device="cuda", generator=g))).contiguous()
        V = torch.rand(r, n, device="cuda", generator=g)
        res = []
        for beside in (False, True):
            ts = []
            for rep in range(5):
                Vc = V.clone()
                torch.cuda.synchronize()
                if beside:
                    with torch.cuda.stream(sstream):
                        for _ in range(3): seng.frob_resid(X, Ut, V0, out=out)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(main)
                st = eng.hals_solve(M, G, Vc, 100, delta=0.0)
                b.record(main)
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            res.append(min(ts[1:]))
        st = st.cpu()
        print(f"  {os.environ.get('NNF_HALS_FORCE', 'default'):8s} r={r:4d} n={n:6d}: alone {res[0]:8.1f} us ({res[0] / 100:5.2f} us/sweep)   "
              f"beside the cost kernel {res[1]:8.1f} us ({res[1] / 100:5.2f} us/sweep)   cnt={int(st[1])} err={int(st[3])} "
              f"sum={float(Vc.double().sum()):.9e}", flush=True)
    sys.exit(0)
specs = sys.argv[1:] or ["50x2000", "30x500", "100x4000", "64x8000", "10x200"]
for layout in ("quad", None):
    env = dict(os.environ)
    env.pop("NNF_HALS_FORCE", None)
    if layout:
        env["NNF_HALS_FORCE"] = layout
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + specs, env=env, check=True)
