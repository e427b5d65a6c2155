cd $GRAFT_REPO_ROOT; O=gpurun_out/r4e; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > $O/pytest.log; tail -2 $O/pytest.log
{
for pm in 100000 1; do echo "aligned alone, ASG_PAIR_MIN_B=$pm"; ASG_ABL_WHAT=aligned ASG_PAIR_MIN_B=$pm timeout 300 python tools/batched_abl.py 512 1024 2048 4096 2>&1 | tail -1; done
for bm in 100000 1; do echo "full alone, ASG_BATCHED_MIN_B=$bm"; ASG_BATCHED_MIN_B=$bm timeout 300 python tools/batched_abl.py 512 1024 2048 4096 2>&1 | tail -1; done
for seq in 0 1; do echo "ASG_BATCHED_SEQ=$seq"; ASG_BATCHED_SEQ=$seq timeout 600 python tools/batched_check.py time 2>&1 | grep -E "B= "; done
} > $O/time.log 2>&1; cat $O/time.log
