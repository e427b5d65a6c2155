#!/bin/bash
# developer build: only asg_fused.hip (NP = 40 only) with extra defines, linked against the objects of the last full build
#   tools/devbuild.sh <out-name> [-DFOO ...]   ->  torch_asg_amd/csrc/variants/lib<out-name>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/torch_asg_amd/csrc; name=$1; shift
mkdir -p $C/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -ffp-contract=off -DASG_DEV_ONLY_NP=${NP:-40} "$@" \
    -c $C/asg_fused.hip -o $C/variants/fused_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/lib$name.so $C/variants/fused_$name.o \
    $C/asg_small_f32.o $C/asg_small_f64.o $C/asg_bwd_f32.o $C/asg_bwd_f64.o $C/asg_generic.o $C/asg_viterbi.o $C/asg_api.o
echo $C/variants/lib$name.so
