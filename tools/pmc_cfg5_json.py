#!/usr/bin/env python3
"""Turn the per-kernel FETCH_SIZE.txt / WRITE_SIZE.txt of tools/pmc_cfg5.sh (directory given as argv[1]) into pmc_cfg5.json there:
raw counter averages, the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE x2 -- checked against the 256 MiB device copy of the
same run -- WRITE_SIZE x1, both in KB), HBM bytes per launch of the large-alphabet kernels.  No GPU needed."""
import json, os, re, sys
R = sys.argv[1]
def rd(name):
    d = {}
    for l in open(os.path.join(R, name + ".txt")):
        m = re.match(r"\S+ (.+?)\s+calls\s+(\d+)\s+avg raw ([\d.]+) KB  min ([\d.]+)", l)
        if m:
            d[m.group(1)] = (float(m.group(3)), float(m.group(4)))
    return d
f, w = rd("FETCH_SIZE"), rd("WRITE_SIZE")
cal = f.get("__amd_rocclr_copyBuffer", (0, 0))[1]
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python tools/pmc_cfg5_probe.py "
                 "(T=60 B=32 N=10000: 59 launches of fwd_step_kernel streaming the same matrices as cfg 5); tools/pmc_cfg5.sh",
       "calibration": "FETCH_SIZE x2 (gfx950 correction of MI355X_MICROARCH.md; the 256 MiB device copy of this run reads raw %.1f KB "
                      "for 262144 KB); WRITE_SIZE x1" % cal}
for k in ("asg::fwd_step_kernel<float", "asg::bwd_gemm_bf3_kernel", "asg::gemm3_pack_kernel", "asg::gemm3_tail_kernel", "asg::bwd_post_kernel<float,"):
    kk = [x for x in f if x.startswith(k)]
    if not kk:
        continue
    fr, wr = f[kk[0]][0], w.get(kk[0], (0, 0))[0]
    out[kk[0]] = {"fetch_raw_kb": fr, "fetch_bytes": fr * 2048, "write_raw_kb": wr, "write_bytes": wr * 1024, "hbm_bytes_per_launch": fr * 2048 + wr * 1024}
st = [x for x in out if "fwd_step_kernel" in x]
if st:
    out[st[0]]["algorithmic_bytes_per_launch"] = 805120000
    out[st[0]]["ratio"] = out[st[0]]["hbm_bytes_per_launch"] / 805120000
    out["dominant_kernel_hbm_bytes_per_launch"] = out[st[0]]["hbm_bytes_per_launch"]
else:
    sys.stderr.write("pmc_cfg5_json: no fwd_step_kernel row in %s/FETCH_SIZE.txt\n" % R)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    out["csrc_sha16"] = bench.csrc_sha16()          # the sources this reading belongs to (bench.py::committed_traffic refuses another tree's)
except Exception as e:
    sys.stderr.write("pmc_cfg5_json: no source fingerprint (%s)\n" % e)
json.dump(out, open(os.path.join(R, "pmc_cfg5.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1800])
sys.exit(0 if st else 1)
