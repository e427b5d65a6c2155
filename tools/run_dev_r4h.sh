cd $GRAFT_REPO_ROOT; O=gpurun_out/r4h; mkdir -p $O
timeout 900 python tools/batch_sweep_fine.py > $O/sweep.txt 2>&1; cat $O/sweep.txt
