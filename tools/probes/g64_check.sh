cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_kernels.py -x -q -k "fp64 or gram_xty" > gpurun_out/g64_a.log 2>&1; tail -3 gpurun_out/g64_a.log
python -m pytest tests/test_gpu_nmf.py tests/test_gpu_fullsize.py -x -q -k "identity or gram or cost" > gpurun_out/g64_b.log 2>&1; tail -3 gpurun_out/g64_b.log
python bench.py --config E --steps 20 --warmup 3 --no-cpu > gpurun_out/bench_E3.log 2>gpurun_out/bench_E3.err; cut -c1-900 gpurun_out/bench_E3.log
python bench.py --steps 20 --warmup 3 --no-cpu --no-extra > gpurun_out/bench_B3.log 2>gpurun_out/bench_B3.err; cut -c1-400 gpurun_out/bench_B3.log
