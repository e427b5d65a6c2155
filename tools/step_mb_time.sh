#!/bin/bash
# developer probe (GPU): large-alphabet shapes with the row blocks per workgroup of the streaming step forced (ASG_STEP_ROW_BLOCKS=5/4/3)
# against the library's own choice
cd "$(dirname "$0")/.."
shapes="${SHAPES:-400,64,1500,30 400,64,2100,30 400,64,2500,30 400,64,3000,30 400,64,3500,30 400,64,4000,30 400,64,5000,30 200,32,7000,30 400,32,3000,30 400,128,3000,30}"
for mb in ${MB:-0 5 4 3}; do
    echo "== ASG_STEP_ROW_BLOCKS=$mb (0: the library's own choice)"
    if [ $mb = 0 ]; then python tools/shape_times.py $shapes 2>/dev/null | grep "T="; else ASG_STEP_ROW_BLOCKS=$mb python tools/shape_times.py $shapes 2>/dev/null | grep "T="; fi
done
