# GPU box: per-kernel average durations (rocprofv3 --stats) of tools/shape_times.py for the shapes given as T,B,N,L
cd /tmp && export TMPDIR=/tmp
for sh in "$@"; do
  rm -rf /tmp/sp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o s -- python $GRAFT_REPO_ROOT/tools/shape_times.py $sh > /tmp/sp.log 2>&1
  grep "us/step" /tmp/sp.log
  python - <<PY
import csv, glob
f = glob.glob("/tmp/sp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("   %-44s calls %4s  avg %9.1f us  %5.1f %%" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
