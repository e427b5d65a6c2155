cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c; mkdir -p $O
export ASG_BATCHED_MIN_B=1
ASG_HIP_LIB=$GRAFT_REPO_ROOT/torch_asg_amd/csrc/variants/libprobe.so timeout 120 python tools/batched_probe.py 512 4096 2>&1 | tail -3 > $O/probe.txt
cat $O/probe.txt
