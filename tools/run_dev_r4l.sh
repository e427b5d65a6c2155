cd $GRAFT_REPO_ROOT; O=gpurun_out/r4l; mkdir -p $O
(timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "viterbi or very_long" 2>&1 | tail -30) > $O/pytest.log; tail -30 $O/pytest.log
