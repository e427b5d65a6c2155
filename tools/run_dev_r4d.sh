cd $GRAFT_REPO_ROOT; O=gpurun_out/r4d; mkdir -p $O
(ASG_PAIR_MIN_B=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest_pairs.log; tail -4 $O/pytest_pairs.log
(ASG_PAIR_MIN_B=1 ASG_BATCHED_MIN_B=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/pytest_both.log; tail -4 $O/pytest_both.log
(ASG_PAIR_MIN_B=1 timeout 600 python tools/batched_check.py check 2>&1 | tail -20) > $O/check.log; tail -5 $O/check.log
for pm in 100000 256; do echo "ASG_PAIR_MIN_B=$pm"; ASG_PAIR_MIN_B=$pm timeout 600 python tools/batched_check.py time 2>&1 | tail -8; done > $O/time.log; cat $O/time.log
