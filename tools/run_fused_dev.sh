# developer run on the GPU box: fused-step correctness on a few shapes, timing, per-role probes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dev
timeout 300 python tools/fused_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dev/check.log
timeout 120 python tools/fused_flags.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dev/flags.log
ASG_DBG=1 ASG_HIP_LIB=$GRAFT_REPO_ROOT/torch_asg_amd/csrc/variants/libasg_probe.so timeout 120 python tools/fused_flags.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dev/probe.log
