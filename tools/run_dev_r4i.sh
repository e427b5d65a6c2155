cd $GRAFT_REPO_ROOT; O=gpurun_out/r4i; mkdir -p $O
(timeout 900 python -m pytest tests/test_hip_variants.py -m gpu -x -q 2>&1 | tail -60) > $O/pytest.log; tail -50 $O/pytest.log
