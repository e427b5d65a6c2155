cd $GRAFT_REPO_ROOT; O=gpurun_out/r4b; mkdir -p $O
(timeout 1200 python tools/batched_check.py all 2>&1 | tail -60) > $O/batched.log; tail -40 $O/batched.log
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/bb
ASG_BATCHED_MIN_B=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bb -o b -- python $GRAFT_REPO_ROOT/tools/bigb_prof.py > /dev/null 2>&1
python3 - <<'PY' > $GRAFT_REPO_ROOT/$O/kernels.txt
import csv, glob, collections
f = glob.glob("/tmp/bb/**/*kernel_trace.csv", recursive=True)[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    per[(n, r.get("Grid_Size_X", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    if "asg" in n: print("%-62s grid %8s calls %3d  min %8.1f med %8.1f max %8.1f us" % (n, g, len(v), min(v), sorted(v)[len(v)//2], max(v)))
PY
cat $GRAFT_REPO_ROOT/$O/kernels.txt
