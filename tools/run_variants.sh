# developer: kernel time of bwd_mfma_kernel / bwd_small_kernel per variant library (rocprofv3 kernel trace)
# usage: bash tools/run_variants.sh "<variant names>" [B ...]
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; V="$1"; shift
cd /tmp && export TMPDIR=/tmp
for v in $V; do
  rm -rf /tmp/pv; mkdir -p /tmp/pv
  L=$R/torch_asg_amd/csrc/variants/lib$v.so; [ "$v" = "main" ] && L=$R/torch_asg_amd/csrc/libasg_hip.so
  ASG_HIP_LIB=$L rocprofv3 --kernel-trace --output-format csv -d /tmp/pv -o t -- python $R/tools/bwd_parts_time.py "$@" > /tmp/pv/out.log 2>&1
  f=$(find /tmp/pv -name "*kernel_trace.csv" | head -1)
  python - "$f" "$v" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "bwd_" not in n and "fwd_" not in n: continue
    k = (n.split("(")[0].split("::")[-1][:28], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))
    d.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
# per (kernel, grid): the sequence is FCC x6, FAC x6, ASG x6 launches -> print min of each third
for k, v in d.items():
    print("%-8s grid %-9s n=%2d  min %.1f  median %.1f us | %s" % (sys.argv[2], k[1], len(v), min(v), sorted(v)[len(v)//2], " ".join("%.0f" % x for x in v)))
PY
done
