# developer run on the GPU box: parity of the fused step on a few shapes, probes, kernel and step times of the dev build
cd $GRAFT_REPO_ROOT; V=$GRAFT_REPO_ROOT/torch_asg_amd/csrc/variants
timeout 300 python tools/fused_check.py 2>&1 | grep -v amdgpu.ids
echo "=== probe"; ASG_DBG=1 ASG_HIP_LIB=$V/libasg_probe.so timeout 120 python tools/fused_flags.py 2>&1 | grep -v amdgpu.ids
echo "=== shipped"; timeout 120 python tools/fused_flags.py 2>&1 | grep -v amdgpu.ids; timeout 120 python tools/step_time.py 2>&1 | tail -1
