cd $GRAFT_REPO_ROOT; O=gpurun_out/r4n; mkdir -p $O
for w in mv1; do
  ASG_HIP_LIB=$GRAFT_REPO_ROOT/torch_asg_amd/csrc/variants/lib$w.so timeout 200 python tools/batched_abl.py 256 512 1024 2048 4096 2>&1 | tail -1
  ASG_HIP_LIB=$GRAFT_REPO_ROOT/torch_asg_amd/csrc/variants/lib$w.so timeout 200 python tools/batch_sweep_fine.py 512 2048 4096 2>&1 | tail -3
done > $O/mv.txt 2>&1; cat $O/mv.txt
