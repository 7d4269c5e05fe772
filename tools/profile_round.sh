cd ${GRAFT_REPO_ROOT:-/root/repo}
for C in B C D E; do bash tools/bench_profile.sh $C all > gpurun_out/prof_$C.log 2>&1; tail -3 gpurun_out/prof_$C.log; done
# one rank's block of config E under the forced sharded protocol on a 1-rank RCCL group
export NNF_BENCH_INIT_PG=1 NNF_BENCH_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
O=gpurun_out/prof_Eblock; mkdir -p $O
timeout -k 10 400 python bench.py --config E --shape 125000,4000,100 --steps 20 --warmup 3 --no-cpu > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python3 $R/bench.py --config E --shape 125000,4000,100 --steps 20 --warmup 3 --no-cpu --no-fixed --no-extra --no-kernels > $R/$O/stats.log 2>&1
cd $R; python tools/prof_summary.py $O/stats > $O/stats_summary.txt; head -14 $O/stats_summary.txt
