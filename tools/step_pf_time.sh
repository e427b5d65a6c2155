#!/bin/bash
# developer probe (GPU): prefetch depth of the streaming step per tile height -- variants built with
#   tools/devbuild_generic.sh pfXYZ -DASG_X_STEP_PF_MB2=X -DASG_X_STEP_PF_MB3=Y -DASG_X_STEP_PF_MB4=Z
# against the shipped library, each with the tile height forced so that every height is timed on the same shapes
cd "$(dirname "$0")/.."
for v in ${VARIANTS:-shipped pf221 pf322 pf433}; do
    if [ $v = shipped ]; then unset ASG_HIP_LIB; else export ASG_HIP_LIB=$PWD/torch_asg_amd/csrc/variants/lib$v.so; fi
    for mb in 2 3 4; do
        echo "== $v, ASG_STEP_ROW_BLOCKS=$mb"
        ASG_STEP_ROW_BLOCKS=$mb python tools/shape_times.py ${SHAPES:-400,64,1100,30 400,64,1500,30 400,32,3000,30 400,64,3000,30 400,32,4000,30} 2>/dev/null | grep "T="
    done
done
