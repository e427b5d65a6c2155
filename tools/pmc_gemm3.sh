# GPU box: L2 counters of the large-alphabet gradient contraction (probe: T=60 B=32 N=10000: K = 1888 rows)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc_gemm3
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  rm -rf /tmp/pq6; timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pq6 -o p -- python $R/tools/pmc_cfg5_probe.py > /tmp/pq6.log 2>&1 || tail -3 /tmp/pq6.log
  python - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/pq6/**/*counter_collection.csv",recursive=True)
if not fs: print("no counters for $c"); raise SystemExit
per=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    n=r["Kernel_Name"]
    if "gemm" in n or "pack" in n:
        key=(r["Counter_Name"], n.replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:40])
        per[key].append(float(r["Counter_Value"]))
for k,v in sorted(per.items()):
    print("%-22s %-42s calls %3d avg %.4g" % (k[0],k[1],len(v),sum(v)/len(v)))
PY
done
