cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dev/bigb; R=$GRAFT_REPO_ROOT
S=${1:-bigb_prof.py}; NL=${2:-40}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/dev/bigb/*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dev/bigb -o bigb -- python $R/tools/$S > $R/gpurun_out/dev/bigb/out.log 2>&1
find $R/gpurun_out/dev/bigb -name "*kernel_trace.csv" | head -1 | xargs -I{} python $R/tools/trace_tail.py {} $NL | grep asg::
