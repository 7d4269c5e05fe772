#!/bin/bash
# one rank's 125000 x 4000 block of config E under the forced sharded protocol (1-rank RCCL group): default switches against the two opt-ins
cd ${GRAFT_REPO_ROOT:-/root/repo}
export NNF_BENCH_INIT_PG=1 NNF_BENCH_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29613 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for SW in "" "NNF_SHARDED_ASYNC=1" "NNF_SHARDED_ASYNC=1 NNF_SHARDED_OVERLAP=1"; do
  echo "== switches: ${SW:-default}"
  env $SW timeout -k 10 300 python bench.py --config E --shape 125000,4000,100 --steps 20 --warmup 3 --no-cpu --no-kernels --no-fixed > gpurun_out/eblock_sw.json 2> gpurun_out/eblock_sw.err || { tail -3 gpurun_out/eblock_sw.err; exit 1; }
  python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/eblock_sw.json") if l.startswith("{")][-1])
print(f"{d['value']:.1f} iterations/s  {d['ms_per_step']:.3f} ms  sweeps mean {d['config'].get('inner_sweeps_mean')}  protocol {d['config'].get('sharded_protocol')}")
PY
done
