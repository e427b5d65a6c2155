cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4o; mkdir -p $O
for mode in streams serial; do
rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o s -- python $R/tools/mode_trace_graph.py $mode > /dev/null 2>&1
python3 - <<PY > $O/graph_trace_$mode.txt
import csv, glob
f = glob.glob("/tmp/st/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-28:]
t0 = int(rows[0]["Start_Timestamp"])
print("launch_mode=$mode, last graph replay (4 steps), us from the first kernel shown")
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    print("%9.1f %9.1f %8.1f  q%-4s %s" % (s, e, e - s, r.get("Queue_Id", "?"), name))
PY
cat $O/graph_trace_$mode.txt
done
