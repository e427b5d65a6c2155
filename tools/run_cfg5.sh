# GPU box: cfg 5 bench line + rocprofv3 kernel stats for profiles/
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/cfg5
timeout 600 python bench.py --config cfg5 > gpurun_out/cfg5/bench_cfg5.json 2> gpurun_out/cfg5/bench.err; tail -c 2500 gpurun_out/cfg5/bench_cfg5.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg5/trace -o cfg5 -- python $R/bench.py --config cfg5 --steps 1 --warmup 1 > $R/gpurun_out/cfg5/under_rocprof.json 2> $R/gpurun_out/cfg5/trace.log
find $R/gpurun_out/cfg5/trace -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-220 | head -14
rm -f $(find $R/gpurun_out/cfg5/trace -name "*kernel_trace.csv")
