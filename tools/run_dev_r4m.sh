cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m; mkdir -p $O; rm -rf /tmp/f64
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/f64 -o f -- python $R/tools/f64_mid_prof.py 128 > /dev/null 2>&1
python3 - <<'PY' > $O/f64_kernels.txt
import csv, glob, collections
f = glob.glob("/tmp/f64/**/*kernel_trace.csv", recursive=True)[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    per[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    if "asg" in n: print("%-72s calls %3d  med %8.1f us  total %9.1f" % (n, len(v), sorted(v)[len(v)//2], sum(v)))
PY
cat $O/f64_kernels.txt
