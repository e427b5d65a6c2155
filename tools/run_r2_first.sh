set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r1/pytest.log
tail -3 gpurun_out/r1/pytest.log
timeout 300 python bench.py --steps 200 --warmup 10 > gpurun_out/r1/bench.json 2> gpurun_out/r1/bench.err; tail -1 gpurun_out/r1/bench.json
timeout 200 python tools/fused_flags.py > gpurun_out/r1/flags.log 2>&1; cat gpurun_out/r1/flags.log
ASG_DBG=1 ASG_HIP_LIB=$GRAFT_REPO_ROOT/torch_asg_amd/csrc/variants/libasg_probe.so timeout 200 python tools/fused_flags.py > gpurun_out/r1/probe.log 2>&1; cat gpurun_out/r1/probe.log
