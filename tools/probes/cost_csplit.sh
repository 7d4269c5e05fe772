#!/bin/bash
# nnf_cost_kernel at config B's shape by number of column splits (NNF_COST_CSPLIT): launch time (kernel trace) and HBM traffic (PMC)
R=${GRAFT_REPO_ROOT:-/root/repo}
for CS in 0 1 2 3; do
  export NNF_COST_CSPLIT=$CS; [ $CS = 0 ] && unset NNF_COST_CSPLIT
  echo "== column splits: ${NNF_COST_CSPLIT:-launch plan}"
  bash $R/tools/probes/cost_traffic.sh 100000x2000x50 | grep "ILi0\|<0" 
  O=$R/gpurun_out/cost_csplit_$CS; (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/probes/cost_only.py 100000x2000x50 > $O.log 2>&1)
  python $R/tools/prof_summary.py $O | grep "nnf_cost_kernel<0"
done
