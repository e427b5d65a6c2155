#!/bin/bash
# Developer builds of the library with ablation bits of the assembly kernel (-DASG_BWD_ABL=n, wrong results: timing only)
# -> torch_asg_amd/csrc/var_libs/libasg_bwdabl<n>.so ; run with ASG_HIP_LIB=<that file>
cd "$(dirname "$0")/.." && mkdir -p torch_asg_amd/csrc/var_libs
for n in "$@"; do
python - <<PY
import sys; sys.path.insert(0, "torch_asg_amd/csrc")
import build
print(build.build(defines=["ASG_BWD_ABL=$n", "ASG_DEV_ONLY_NP=40"], out="torch_asg_amd/csrc/var_libs/libasg_bwdabl$n.so"))
PY
done
