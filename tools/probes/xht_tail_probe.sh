#!/bin/bash
# X H^T at config B's shape with the k-split tail (the 106 row tiles beyond a whole round shared by all workgroups) against the
# (4,3)-tiles-per-wave plan (NNF_XHT_TAIL=0): launch time + reduction, result against float64, then the bench line of either.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for T in 1 0; do
  NNF_XHT_TAIL=$T python tools/probes/xht_nt2_probe.py 100000x2000x50 100000x2000x64 98000x2000x50 110000x2000x40 2>&1 | grep "rank" | sed "s/^NNF_XHT_NT2=0/NNF_XHT_TAIL=$T/"
done
for T in 1 0 1 0; do
  NNF_XHT_TAIL=$T python bench.py --steps 20 --warmup 3 --no-cpu --no-extra --no-fixed > gpurun_out/bench_tail_$T.log 2>gpurun_out/bench_tail_$T.err
  python - $T <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/bench_tail_{sys.argv[1]}.log") if l.startswith("{")][-1])
x = [e for e in d["roofline_more"] if "xht" in e["kernel"]][0]
print(f"NNF_XHT_TAIL={sys.argv[1]}: {d['value']:.1f} iterations/s, X H^T in the loop {x['launch_ms'] * 1e3:.1f} us ({x['frac']:.3f}), stand-alone {x['standalone']['launch_ms'] * 1e3:.1f}")
PY
done
