# GPU box: FETCH_SIZE / WRITE_SIZE of the fused kernels (separate passes), printed per kernel
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pq; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pq -o p -- python $R/tools/pmc_probe.py > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pq/**/*counter_collection.csv",recursive=True)[0]
per=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="$c" and "fused" in r["Kernel_Name"]: per[r["Kernel_Name"][31:46]].append(float(r["Counter_Value"]))
for k,v in per.items(): print("$c", k, "%.0f KB raw (x2 for FETCH)" % (sum(v)/len(v)))
PY
done
