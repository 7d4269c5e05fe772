#!/bin/bash
# Timing-only ablations of nnf_xht_kernel (k_stream.hip, XHT_ABL) at config B's shape: which part of the kernel its time belongs to.
#   here (no GPU):   bash tools/xht_ablate.sh build     -> tools/abl/libnnfac_xht{1..5}.so
#   on the GPU box:  bash tools/xht_ablate.sh run       -> gpurun_out/abl_xht.txt   (results of the ablated builds are WRONG by design)
# 1: no X stream   2: no MFMA   3: no leftover-rank FMAs   4: no LDS restage / barrier   5: no ragged-tail branch
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/nn_fac_amd/csrc
mkdir -p $R/tools/abl
if [ "$1" = "build" ]; then
  OBJS=$(ls $C/build/*.o | grep -v k_stream.o)
  for v in 1 2 3 4 5; do
    /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -DXHT_ABL=$v -c $C/k_stream.hip -o $R/tools/abl/k_stream_abl$v.o &
  done
  wait
  for v in 1 2 3 4 5; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/abl/libnnfac_xht$v.so $OBJS $R/tools/abl/k_stream_abl$v.o -ldl
    rm -f $R/tools/abl/k_stream_abl$v.o
  done
  exit 0
fi
cp $R/nn_fac_amd/libnnfac_hip.so /tmp/libnnfac_keep.so
: > $R/gpurun_out/abl_xht.txt
for v in 0 1 2 3 4 5; do
  if [ $v = 0 ]; then cp /tmp/libnnfac_keep.so $R/nn_fac_amd/libnnfac_hip.so; else cp $R/tools/abl/libnnfac_xht$v.so $R/nn_fac_amd/libnnfac_hip.so; fi
  python - $v <<'PY' >> $R/gpurun_out/abl_xht.txt 2>&1
import sys, torch
sys.path.insert(0, ".")
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
m, n, r = 100000, 2000, 50
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.rand(m, n, device="cuda", generator=g)
V = torch.rand(r, n, device="cuda", generator=g)
for _ in range(3): eng.xht(X, V)
ms = min(eng.time_kernel("xht", lambda: eng.xht(X, V)) for _ in range(3))
print(f"ABL={sys.argv[1]}  X H^T main kernel {1e3 * ms:.1f} us")
PY
done
cp /tmp/libnnfac_keep.so $R/nn_fac_amd/libnnfac_hip.so
grep ABL $R/gpurun_out/abl_xht.txt
