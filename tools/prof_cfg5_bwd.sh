# GPU box: rocprofv3 kernel stats of one cfg-5 forward + backward (tools/cfg5_bwd_time.py) -> gpurun_out/cfg5_bwd_stats.csv
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/pb5; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb5 -o p -- python $R/tools/cfg5_bwd_time.py > $R/gpurun_out/cfg5_bwd_prof.log 2>&1
f=$(find /tmp/pb5 -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/cfg5_bwd_stats.csv; cut -c1-200 $f | head -16
