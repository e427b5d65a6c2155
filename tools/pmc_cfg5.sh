# GPU box: FETCH_SIZE / WRITE_SIZE of the large-alphabet kernels (separate passes; counters only with --kernel-trace) ->
# gpurun_out/pmc_cfg5/{FETCH_SIZE,WRITE_SIZE}.txt and gpurun_out/pmc_cfg5/pmc_cfg5.json (copy to profiles/<tag>_pmc_cfg5.json)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc_cfg5
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pq5; timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pq5 -o p -- python $R/tools/pmc_cfg5_probe.py > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pq5/**/*counter_collection.csv",recursive=True)[0]
per=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="$c":
        n=r["Kernel_Name"]
        key=n.replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:44]
        per[key].append(float(r["Counter_Value"]))
out=open("$R/gpurun_out/pmc_cfg5/$c.txt","w")
for k,v in sorted(per.items(), key=lambda kv:-sum(kv[1])):
    line="$c %-46s calls %4d  avg raw %.1f KB  min %.1f  max %.1f" % (k,len(v),sum(v)/len(v),min(v),max(v))
    print(line); out.write(line+"\n")
PY
done
python $R/tools/pmc_cfg5_json.py $R/gpurun_out/pmc_cfg5
