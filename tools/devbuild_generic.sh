#!/bin/bash
# developer build: the generic path's translation units (asg_generic*.hip) with extra defines, linked against the objects of the last full build
#   tools/devbuild_generic.sh <name> [-D...]   ->  torch_asg_amd/csrc/variants/lib<name>.so   (ASG_HIP_LIB=... to use it)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/torch_asg_amd/csrc; name=$1; shift
mkdir -p $C/variants
objs=""
for tu in asg_generic asg_generic_step asg_generic_aligned asg_generic_grad; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Wno-unused-value -Wno-inline-asm -ffp-contract=off "$@" -c $C/$tu.hip -o $C/variants/${tu}_$name.o &
  objs="$objs $C/variants/${tu}_$name.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/lib$name.so $objs \
    $C/asg_small_f32.o $C/asg_small_f64.o $C/asg_bwd_f32.o $C/asg_bwd_f64.o $C/asg_fused.o $C/asg_viterbi.o $C/asg_api.o
echo $C/variants/lib$name.so
