# GPU box: instruction-issue counters of the stand-alone route's kernels at T=400 B=512 N=40 (one SQ pass, --kernel-trace only).
# Question (round 6): are the recursion chains at B = 512 latency-bound with idle issue slots (then assembly work
# could hide behind them) or is the SIMD's issue port busy?  SQ_* count quad-cycles summed over wavefronts (MI355X_MICROARCH.md).
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc_issue
rm -rf /tmp/pqi; timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d /tmp/pqi -o p -- python $R/tools/pmc_standalone_probe.py > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pqi/**/*counter_collection.csv",recursive=True)[0]
per=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:44]
    per[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out=open("$R/gpurun_out/pmc_issue/issue.txt","w")
def p(s):
    print(s); out.write(s+"\n")
p("per launch, averaged over calls; SQ_* in quad-cycles summed over all wavefronts of the launch")
for k,c in per.items():
    if not k.startswith("asg::"): continue
    a={n:sum(v)/len(v) for n,v in c.items()}
    wc=a.get("SQ_WAVE_CYCLES",0) or 1
    p("%-44s waves %8.0f  wave_cycles %12.0f  active_any %5.1f %%  active_valu %5.1f %%  wait_any %5.1f %%  wait_inst_any %5.1f %%  valu insts/wave %8.0f  busy_cycles %10.0f"
      % (k,a.get("SQ_WAVES",0),wc,100*a.get("SQ_ACTIVE_INST_ANY",0)/wc,100*a.get("SQ_ACTIVE_INST_VALU",0)/wc,100*a.get("SQ_WAIT_ANY",0)/wc,
         100*a.get("SQ_WAIT_INST_ANY",0)/wc,a.get("SQ_INSTS_VALU",0)/max(a.get("SQ_WAVES",1),1),a.get("SQ_BUSY_CYCLES",0)))
PY
