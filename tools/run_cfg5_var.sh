# GPU box: forward time of the large-alphabet path at cfg 5 for the variant libraries named in LIBS (default: the shipped one)
cd $GRAFT_REPO_ROOT; V=$GRAFT_REPO_ROOT/torch_asg_amd/csrc/variants
echo "shipped:"; timeout 100 python tools/cfg5_fwd_time.py 2>&1 | tail -1
echo "shipped, one cooperative launch:"; ASG_PERSIST=1 timeout 100 python tools/cfg5_fwd_time.py 2>&1 | tail -1
for L in $LIBS; do echo "$L:"; ASG_HIP_LIB=$V/lib$L.so timeout 100 python tools/cfg5_fwd_time.py 2>&1 | tail -1; done
