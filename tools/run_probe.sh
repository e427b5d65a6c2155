cd /tmp && export TMPDIR=/tmp
ASG_DBG=1 ASG_HIP_LIB=$GRAFT_REPO_ROOT/var_libs/libasg_probe.so timeout 300 python $GRAFT_REPO_ROOT/tools/fused_flags.py 2>&1 | grep -E "phases|forward"
timeout 300 python $GRAFT_REPO_ROOT/tools/fused_flags.py 2>&1 | grep -E "forward"
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_f -o f -- python $GRAFT_REPO_ROOT/tools/fused_flags.py > /dev/null 2>&1
head -8 $GRAFT_REPO_ROOT/gpurun_out/prof_f/*/f_kernel_stats.csv | cut -c1-200
