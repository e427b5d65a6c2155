#!/bin/bash
# developer probe (GPU): large-alphabet shapes at B >= 64, two batch tiles per workgroup (default) against one (ASG_STEP_ONE_TILE=1)
cd "$(dirname "$0")/.."
shapes="${@:-400,64,1500,30 400,64,3000,30 400,64,5000,30 400,128,3000,30 200,96,5000,30 400,32,3000,30}"
echo "== two batch tiles per workgroup (default)"; python tools/shape_times.py $shapes
echo "== one batch tile per workgroup"; ASG_STEP_ONE_TILE=1 python tools/shape_times.py $shapes
