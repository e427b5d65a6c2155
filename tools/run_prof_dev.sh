cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dev/prof; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dev/prof -o bench -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/dev/prof/bench.json 2> $R/gpurun_out/dev/prof/trace.log
cat $R/gpurun_out/dev/prof/bench.json | head -c 600; echo
find $R/gpurun_out/dev/prof -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -20
