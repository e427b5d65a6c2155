#!/bin/bash
# Developer run (GPU box): duration of the assembly kernel per ablation build (tools/devbuild_bwd.sh) under rocprofv3
R=$(pwd); O=$R/gpurun_out/${1:-abl}; mkdir -p $O; shift
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "0" ]; then unset ASG_HIP_LIB; else export ASG_HIP_LIB=$R/torch_asg_amd/csrc/var_libs/libasg_bwdabl$v.so; fi
  export ASG_NO_BINDING=1
  rm -rf $O/p$v; timeout 300 rocprofv3 --kernel-trace -d $O/p$v -o t -- python $R/tools/bigb_prof.py > $O/p$v.log 2>&1
  echo "== variant $v"; python $R/tools/kernel_medians.py $O/p$v bwd_mfma
done
