# GPU box: instruction-cache counters of the fused forward kernel for the libraries named in LIBS (variants/lib<name>.so)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; V=$R/torch_asg_amd/csrc/variants
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_INST_CYCLES_VMEM[A-Z_]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_IFETCH_LEVEL" | sort -u | tr '\n' ' '; echo
for L in ${LIBS:-base asg_dev}; do
  for c in "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL" "SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pq; ASG_HIP_LIB=$V/lib$L.so rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pq -o p -- python $R/tools/pmc_probe.py > /tmp/pq.log 2>&1
  python - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/pq/**/*counter_collection.csv",recursive=True)
if not fs: print("$L: no counter file"); print(open("/tmp/pq.log").read()[-600:])
else:
    per=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "fused_fwd" in r["Kernel_Name"]: per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("$L", " ".join("%s %.0f" % (k, sum(v)/len(v)) for k,v in sorted(per.items())))
PY
  done
done
