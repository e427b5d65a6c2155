# developer A/B on the GPU box: base (last commit) vs the dev build of asg_fused.hip: probes, kernel times, step time
cd $GRAFT_REPO_ROOT; V=$GRAFT_REPO_ROOT/torch_asg_amd/csrc/variants
for L in ${LIBS:-base_probe asg_probe}; do echo "=== $L"; ASG_DBG=1 ASG_HIP_LIB=$V/lib$L.so timeout 120 python tools/fused_flags.py 2>&1 | grep -v amdgpu.ids; done
for L in ${TLIBS:-base asg_dev}; do echo "=== $L"; ASG_HIP_LIB=$V/lib$L.so timeout 120 python tools/fused_flags.py 2>&1 | grep -v amdgpu.ids; ASG_HIP_LIB=$V/lib$L.so timeout 120 python tools/step_time.py 2>&1 | tail -1; done
