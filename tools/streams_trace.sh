# GPU box: rocprofv3 kernel trace of ONE cfg-3 step in launch_mode='streams' (eager): the full-lattice chains on the
# caller's stream, the aligned chains on the side stream -- start / end / queue of every kernel of the last step, and how
# long the two streams' recursion kernels ran at the same time.  -> gpurun_out/streams/streams_step_trace.txt
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/streams; mkdir -p $O; rm -rf /tmp/st
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o s -- python $R/tools/mode_trace.py streams > /dev/null 2> $O/trace.log
python - <<PY > $O/streams_step_trace.txt
import csv, glob
f = glob.glob("/tmp/st/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "asg::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step = the kernels after the last fwd recursion launch on the caller's stream
starts = [i for i, r in enumerate(rows) if "fwd_duo_kernel" in r["Kernel_Name"] or "fwd_small_kernel" in r["Kernel_Name"]]
# two recursion launches per step (full lattice, aligned lattice): take the last pair and everything after it
first = starts[-2]
step = rows[first:]
t0 = int(step[0]["Start_Timestamp"])
print("launch_mode='streams', cfg 3 (T=400 B=64 N=40 L=30 fp32), eager, last of 6 steps; times in us from the first kernel of the step")
print("%9s %9s %8s  %-10s %s" % ("start", "end", "dur", "queue", "kernel"))
rec = []
for r in step:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    q = r.get("Queue_Id", r.get("Stream_Id", "?"))
    print("%9.1f %9.1f %8.1f  %-10s %s  grid %sx%s" % (s, e, e - s, q, name[:70], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "")))
    if "fwd_duo_kernel" in name or "fwd_small_kernel" in name: rec.append((s, e, q))
if len(rec) >= 2:
    (s0, e0, q0), (s1, e1, q1) = rec[0], rec[1]
    ov = max(0.0, min(e0, e1) - max(s0, s1))
    print("recursion kernels: %.1f us on queue %s and %.1f us on queue %s, %.1f us of them at the same time (%.0f %% of the shorter one); both done %.1f us after the first started"
          % (e0 - s0, q0, e1 - s1, q1, ov, 100 * ov / min(e0 - s0, e1 - s1), max(e0, e1) - min(s0, s1)))
PY
cat $O/streams_step_trace.txt
