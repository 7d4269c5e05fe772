cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_bigrank.py -x -q -k "xht or gram_xty or config_e or identity or above_rank" > gpurun_out/nt2_tests.log 2>&1; tail -3 gpurun_out/nt2_tests.log | cut -c1-200
python bench.py --config E --steps 20 --warmup 3 --no-cpu --no-fixed > gpurun_out/bench_E5.log 2>gpurun_out/bench_E5.err; cut -c1-330 gpurun_out/bench_E5.log
