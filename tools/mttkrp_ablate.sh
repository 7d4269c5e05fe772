#!/bin/bash
# Timing-only ablations of nnf_mttkrp_rows_kernel (k_mttkrp.hip, MTTKRP_ABL): which part of the kernel its time belongs to.
#   here (no GPU):   bash tools/mttkrp_ablate.sh build [seg]   -> tools/abl/libnnfac_abl{1,2,3,4,5}.so
#   on the GPU box:  bash tools/mttkrp_ablate.sh run [seg]     -> gpurun_out/abl.txt   (results of the ablated builds are WRONG by design)
# 1: no Khatri-Rao generation   2: no MFMA   3: no X stream   4: no LDS restage / barrier   5 (seg only): no ragged-tail branch
# `seg`: the segment kernel of modes 0 / 1 (-DSEG_ABL) instead of the row kernel of mode 2 (-DMTTKRP_ABL)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/nn_fac_amd/csrc
KERN=${2:-rows}; if [ "$KERN" = seg ]; then DEF=SEG_ABL; MODE=0; VS="1 2 3 4 5"; else DEF=MTTKRP_ABL; MODE=2; VS="1 2 3 4"; fi
mkdir -p $R/tools/abl
if [ "$1" = "build" ]; then
  OBJS=$(ls $C/build/*.o | grep -v k_mttkrp.o)
  for v in $VS; do
    /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -D$DEF=$v -c $C/k_mttkrp.hip -o $R/tools/abl/k_mttkrp_abl$v.o &
  done
  wait
  for v in $VS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/abl/libnnfac_abl$v.so $OBJS $R/tools/abl/k_mttkrp_abl$v.o -ldl
    rm -f $R/tools/abl/k_mttkrp_abl$v.o
  done
  exit 0
fi
cp $R/nn_fac_amd/libnnfac_hip.so /tmp/libnnfac_keep.so
: > $R/gpurun_out/abl.txt
for v in 0 $VS; do
  if [ $v = 0 ]; then cp /tmp/libnnfac_keep.so $R/nn_fac_amd/libnnfac_hip.so; else cp $R/tools/abl/libnnfac_abl$v.so $R/nn_fac_amd/libnnfac_hip.so; fi
  python - $v $MODE <<'PY' >> $R/gpurun_out/abl.txt 2>&1
import sys, torch
sys.path.insert(0, ".")
from nn_fac_amd.engine import get_engine
eng = get_engine("cuda:0")
I, R = 500, 30
g = torch.Generator(device="cuda").manual_seed(1)
T = torch.rand(I, I, I, device="cuda", generator=g)
Ft = [torch.rand(R, I, device="cuda", generator=g) for _ in range(3)]
mode = int(sys.argv[2])
for _ in range(3): eng.mttkrp3(T, Ft, mode)
ms = min(eng.time_kernel("mttkrp", lambda: eng.mttkrp3(T, Ft, mode)) for _ in range(3))
print(f"ABL={sys.argv[1]}  mode-{mode} MTTKRP main kernel {1e3 * ms:.1f} us")
PY
done
cp /tmp/libnnfac_keep.so $R/nn_fac_amd/libnnfac_hip.so
cat $R/gpurun_out/abl.txt
