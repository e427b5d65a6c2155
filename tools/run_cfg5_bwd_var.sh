# GPU box: backward time of the large-alphabet path at cfg 5 (the gradient contraction) for the shipped library and the variants in LIBS
cd $GRAFT_REPO_ROOT; V=$GRAFT_REPO_ROOT/torch_asg_amd/csrc/variants
for rep in 1 2; do
echo "shipped:"; timeout 120 python tools/cfg5_bwd_time.py 2>&1 | tail -1
for L in $LIBS; do echo "$L:"; ASG_HIP_LIB=$V/lib$L.so timeout 120 python tools/cfg5_bwd_time.py 2>&1 | tail -1; done
done
