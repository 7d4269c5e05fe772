#!/bin/bash
# Compiles ONE instantiation of the fused MU kernels alone and prints its registers / spills / scratch; ISA left in $OUT.
#   bash tools/mu_kernel_regs.sh left|right MT REM BM [VEC=true]      (BM: 1 KL, 2 FROB, 3 KL+cost, 9 general beta)
SIDE=${1:-left}; MT=${2:-4}; REM=${3:-0}; BM=${4:-1}; VEC=${5:-true}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/mu_dev_${SIDE}_${MT}_${REM}_${BM}}
mkdir -p $OUT
if [ "$SIDE" = left ]; then
  SIG='(const float*, int64_t, int64_t, int64_t, const float*, int64_t, const float*, int64_t, int, float, const double*, float, float*, int64_t, int, int, int, mu_left_extra)'
else
  SIG='(const float*, int64_t, int64_t, int64_t, const float*, int64_t, const float*, int64_t, int, float, float*, float*, int64_t, int, int, int64_t, int)'
fi
cat > $OUT/dev.hip <<EOT
#include "k_mu_kernels.h"
template __global__ void nnf_mu_${SIDE}_kernel<$MT, $REM, $BM, $VEC>$SIG;
EOT
cd $OUT && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -I$R/nn_fac_amd/csrc -I$R/include -save-temps=obj $EXTRA -c dev.hip -o dev.o || exit 1
grep -E "^\s+\.(vgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):" dev-hip-amdgcn-amd-amdhsa-gfx950.s | tr -s ' ' | tr '\n' ' '; echo
echo "ISA: $OUT/dev-hip-amdgcn-amd-amdhsa-gfx950.s"
