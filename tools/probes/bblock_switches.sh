#!/bin/bash
# config B's block under the forced sharded protocol on one rank (1-rank RCCL group): unsharded, default protocol, and the opt-in switches
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --steps 40 --warmup 5 --no-cpu --no-kernels --no-fixed --no-extra > gpurun_out/bblock_sw.json 2> gpurun_out/bblock_sw.err || { tail -3 gpurun_out/bblock_sw.err; exit 1; }
  python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bblock_sw.json") if l.startswith("{")][-1])
print(f"{d['value']:.1f} iterations/s  {d['ms_per_step']:.3f} ms  protocol {d['config'].get('sharded_protocol')}")
PY
}
echo "== unsharded"; run
export NNF_BENCH_INIT_PG=1 NNF_BENCH_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29615 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
echo "== sharded protocol, default"; run
echo "== NNF_SHARDED_ASYNC=1"; NNF_SHARDED_ASYNC=1 run
echo "== NNF_SHARDED_ASYNC=1 NNF_SHARDED_OVERLAP=1"; NNF_SHARDED_ASYNC=1 NNF_SHARDED_OVERLAP=1 run
