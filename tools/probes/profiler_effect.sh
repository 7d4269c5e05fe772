cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/probes/profiler_effect_probe.py > gpurun_out/pe_plain.log 2>&1 && cat gpurun_out/pe_plain.log
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pe_prof -- python3 $R/tools/probes/profiler_effect_probe.py > $R/gpurun_out/pe_prof.log 2>&1
cd $R; grep -v "^[EW]2026" gpurun_out/pe_prof.log | tail -5; python tools/prof_summary.py gpurun_out/pe_prof | grep "nnf_mu\|nnf_xty" 
