cd $GRAFT_REPO_ROOT; O=gpurun_out/r4f; mkdir -p $O
(ASG_PAIR_MIN_B=2048 timeout 600 python tools/batched_check.py check 2>&1 | tail -20) > $O/check.log; tail -5 $O/check.log
{
for sp in 0 1; do echo "full alone, batched, ASG_X16_SPLIT=$sp"; ASG_X16_SPLIT=$sp ASG_BATCHED_MIN_B=1 timeout 300 python tools/batched_abl.py 512 1024 2048 4096 8192 2>&1 | tail -1; done
for seq in 0 1; do echo "ASG_BATCHED_SEQ=$seq ASG_PAIR_MIN_B=2048"; ASG_PAIR_MIN_B=2048 ASG_BATCHED_SEQ=$seq timeout 600 python tools/batched_check.py time 2>&1 | grep -E "B= "; done
} > $O/time.log 2>&1; cat $O/time.log
