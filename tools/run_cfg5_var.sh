# GPU box: forward time of the large-alphabet path at cfg 5 for the variant libraries named in LIBS (default: the shipped one),
# twice round-robin (the first process on a fresh box runs slow: clocks / page-in)
cd $GRAFT_REPO_ROOT; V=$GRAFT_REPO_ROOT/torch_asg_amd/csrc/variants
timeout 25 python tools/cfg5_fwd_time.py > /dev/null 2>&1
for rep in 1 2; do
echo "shipped:"; timeout 25 python tools/cfg5_fwd_time.py 2>&1 | tail -1
for L in $LIBS; do echo "$L:"; ASG_HIP_LIB=$V/lib$L.so timeout 25 python tools/cfg5_fwd_time.py 2>&1 | tail -1; done
done
