#!/bin/bash
# Developer run (GPU box): parity of the stand-alone routes, then the step time over the batch size and the assembly kernel's time.
R=$(pwd); O=$R/gpurun_out/${1:-r4q}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_batched.py -m gpu -x -q -k "standalone or batched or reduction or serial or streams" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python tools/batch_sweep_fine.py 96 128 192 256 512 1024 2048 4096 > $O/sweep.txt 2>&1
cat $O/sweep.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bigb -- python $R/tools/bigb_prof.py > $O/prof.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
