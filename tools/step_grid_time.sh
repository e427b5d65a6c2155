#!/bin/bash
# developer probe (GPU, -DASG_DEV_PROBES build in variants/libprobes.so): step time of large-alphabet shapes over a grid of forced tile
# heights (ASG_STEP_ROW_BLOCKS) and K-slice counts (ASG_STEP_KS), and the library's own choice (printed by ASG_STEP_SHOW)
cd "$(dirname "$0")/.."
export ASG_HIP_LIB=$PWD/torch_asg_amd/csrc/variants/libprobes.so
for shape in ${SHAPES:-400,64,1500,30 400,64,2100,30 400,64,2500,30 400,16,2100,30 400,32,3000,30 400,32,5000,30}; do
    echo "== $shape"
    ASG_STEP_SHOW=1 python tools/shape_times.py $shape 2>&1 | grep "T=\|step grid" | sort -u | sed 's/^/   own: /'
    for mb in ${MBS:-5 4 3 2}; do for ks in ${KS:-1 2 3 4}; do
        echo -n "   mb=$mb ks=$ks: "; ASG_STEP_ROW_BLOCKS=$mb ASG_STEP_KS=$ks python tools/shape_times.py $shape 2>/dev/null | grep "T=" | sed 's/.*: //'
    done; done
done
