#!/bin/bash
# developer probe (GPU, ASG_DEV_PROBES build in variants/libprobes.so): step time of large-alphabet shapes with a forced number of K slices
cd "$(dirname "$0")/.."
export ASG_HIP_LIB=$PWD/torch_asg_amd/csrc/variants/libprobes.so
shapes="${SHAPES:-400,64,3000,30 400,64,2200,30 400,32,5000,30 400,64,5000,30}"
for ks in ${KS:-0 1 2 3 4 5 6 8}; do
    echo "== ASG_STEP_KS=$ks (0: the library's own choice)"
    if [ $ks = 0 ]; then python tools/shape_times.py $shapes 2>/dev/null; else ASG_STEP_KS=$ks python tools/shape_times.py $shapes 2>/dev/null; fi
done
