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
python - <<PY
import json,re
R="$R/gpurun_out/pmc_cfg5/"
def rd(name):
    d={}
    for l in open(R+name+".txt"):
        m=re.match(r"\S+ (\S+)\s+calls\s+(\d+)\s+avg raw ([\d.]+) KB  min ([\d.]+)",l)
        if m: d[m.group(1)]=(float(m.group(3)),float(m.group(4)))
    return d
f,w=rd("FETCH_SIZE"),rd("WRITE_SIZE")
cal=f.get("__amd_rocclr_copyBuffer",(0,0))[1]
out={"source":"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python tools/pmc_cfg5_probe.py (T=60 B=32 N=10000: 59 launches of fwd_step_kernel streaming the same matrices as cfg 5); tools/pmc_cfg5.sh",
     "calibration":"FETCH_SIZE x2 (gfx950 correction of MI355X_MICROARCH.md; the 256 MiB device copy of this run reads raw %.1f KB for 262144 KB); WRITE_SIZE x1" % cal}
for k in ("asg::fwd_step_kernel<float>","asg::bwd_gemm_bf3_kernel","asg::gemm3_pack_kernel","asg::bwd_post_kernel<float,"):
    kk=[x for x in f if x.startswith(k[:28])]
    if not kk: continue
    fr,wr=f[kk[0]][0],w.get(kk[0],(0,0))[0]
    out[kk[0]]={"fetch_raw_kb":fr,"fetch_bytes":fr*2048,"write_raw_kb":wr,"write_bytes":wr*1024,"hbm_bytes_per_launch":fr*2048+wr*1024}
st=[x for x in out if "fwd_step_kernel" in x]
if st:
    out[st[0]]["algorithmic_bytes_per_launch"]=805120000
    out[st[0]]["ratio"]=out[st[0]]["hbm_bytes_per_launch"]/805120000
    out["dominant_kernel_hbm_bytes_per_launch"]=out[st[0]]["hbm_bytes_per_launch"]
json.dump(out,open(R+"pmc_cfg5.json","w"),indent=1)
print(json.dumps(out,indent=1)[:1500])
PY
