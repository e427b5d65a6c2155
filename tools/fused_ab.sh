#!/bin/bash
# Developer run (GPU box): cfg-3 bench line (no extras) under rocprofv3 for the shipped library and for the libraries named
# on the command line (paths under torch_asg_amd/csrc/var_libs/), alternating; prints the fused kernels' medians and ms_per_step.
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for i in 1 2; do
  for v in shipped "$@"; do
    if [ "$v" = shipped ]; then unset ASG_HIP_LIB; unset ASG_NO_BINDING; else export ASG_HIP_LIB=$R/torch_asg_amd/csrc/var_libs/$v ASG_NO_BINDING=1; fi
    rm -rf /tmp/pb; timeout 300 rocprofv3 --kernel-trace -d /tmp/pb -o t -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra > /tmp/b.json 2>/tmp/b.err
    echo "== $v: $(python -c 'import json;d=json.load(open("/tmp/b.json"));print("%.2f us/step (profiled)" % (1e3*d["ms_per_step"]))')"
    python $R/tools/kernel_medians.py /tmp/pb fused_
    timeout 200 python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | python -c 'import json,sys;d=json.loads(sys.stdin.read());print("   unprofiled %.2f us/step" % (1e3*d["ms_per_step"]))'
  done
done
