#!/bin/bash
# Residual-refresh interval of the wave-per-column solve (k_hals_wave.hip: 8 sweeps): accuracy on the ill-conditioned cases of
# tools/stress_parity.py (seeds 5, 6: kappa > 1e7) and us per sweep (tools/probes/vside_probe.py), builds with 8 / 4 / 2 / 1.
#   bash tools/abl_build.sh wave_refN k_hals_wave.hip k_hals_wave.o -DWAVE_REFRESH_V=N   (N = 4, 2, 1), then this script on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for N in 8 4 2 1; do
  if [ $N = 8 ]; then unset NNF_LIBRARY; else export NNF_LIBRARY=$R/tools/abl/libnnfac_wave_ref$N.so; fi
  echo "== refresh every $N sweeps"
  for seed in 5 6; do timeout -k 10 300 python tools/stress_parity.py $seed 60 v 2>&1 | grep "NOTE\|CASE\|stress seed" | tail -8; done
  timeout -k 10 200 python tools/probes/vside_probe.py 50x2000 100x4000 30x500 2>&1 | grep -v amdgpu | tail -4
done
