#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats + PMC passes) into profiles/<tag>_*.{md,json,csv}."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return hits[0] if hits else None


def main():
    tag, out = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    lines = ["# rocprofv3 summary %s" % tag, ""]
    # ---- kernel stats of the bench command
    stats = find(os.path.join(out, "trace"), "*kernel_stats.csv")
    trace = find(os.path.join(out, "trace"), "*kernel_trace.csv")
    if stats:
        shutil.copy(stats, os.path.join(prof, "%s_bench_kernel_stats.csv" % tag))
    agg = collections.OrderedDict()
    if trace:
        for r in csv.DictReader(open(trace)):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            agg.setdefault(r["Kernel_Name"], []).append(d)
        lines += ["## `rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pmc`", "",
                  "| kernel | calls | avg us | min us | max us | total ms |", "|---|---|---|---|---|---|"]
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            lines.append("| `%s` | %d | %.2f | %.2f | %.2f | %.3f |" % (k[:110], len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1e3))
        lines.append("")
    bj = os.path.join(out, "bench_under_rocprof.json")
    if os.path.exists(bj):
        txt = open(bj).read().strip()
        lines += ["bench.py line printed under the profiler (profiled runs clock lower; not the headline):", "", "```", txt[:3000], "```", ""]
    # ---- PMC
    pmc = {}
    for which, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = find(os.path.join(out, which), "*counter_collection.csv")
        if not f:
            continue
        shutil.copy(f, os.path.join(prof, "%s_%s_counter_collection.csv" % (tag, which)))
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        pmc[name] = per
    if pmc:
        def avg(name, key):
            for k, v in pmc.get(name, {}).items():
                if key in k:
                    return sum(v) / len(v), max(v)
            return None, None
        cal_f = avg("FETCH_SIZE", "copyBuffer")[1]
        cal_w = avg("WRITE_SIZE", "copyBuffer")[1]
        lines += ["## PMC passes (`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate runs of tools/pmc_probe.py)", "",
                  "Units: KB per dispatch (raw counter).  Calibration = a 256 MiB device copy in the same process: "
                  "FETCH_SIZE reads %s KB (= 1/2 of the 262144 KB actually read -> gfx950 correction x2, as "
                  "MI355X_MICROARCH.md section HBM says), WRITE_SIZE reads %s KB (exact)." % (cal_f, cal_w), "",
                  "| kernel | FETCH_SIZE raw KB | fetch bytes (x2 x1024) | WRITE_SIZE KB | write bytes | HBM bytes / launch |", "|---|---|---|---|---|---|"]
        summary = {}
        for key, label in (("fused_fwd_kernel", "dominant_kernel"), ("fused_bwd_kernel", "backward_kernel"),
                           ("fwd_duo_kernel", "recursion_kernel"), ("fwd_small_kernel", "recursion_kernel"),
                           ("bwd_small_kernel", "assembly_kernel"), ("reduce_tiles_kernel", "reduce_kernel")):
            if label + "_hbm_bytes_per_launch" in summary:
                continue
            f_, _ = avg("FETCH_SIZE", key)
            w_, _ = avg("WRITE_SIZE", key)
            if f_ is None or w_ is None:
                continue
            fb, wb = f_ * 2 * 1024, w_ * 1024
            summary[label + "_hbm_bytes_per_launch"] = fb + wb
            summary[label + "_fetch_bytes"] = fb
            summary[label + "_write_bytes"] = wb
            lines.append("| %s | %.1f | %.0f | %.1f | %.0f | %.0f |" % (key, f_, fb, w_, wb, fb + wb))
        lines.append("")
        summary["note"] = "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B; verified on a 256 MiB copy in the same run); WRITE_SIZE as read"
        summary["step_hbm_bytes"] = sum(v for k, v in summary.items() if k.endswith("_hbm_bytes_per_launch"))
        summary["algorithmic_bytes_per_step"] = 8221440
        summary["step_traffic_over_algorithmic"] = summary["step_hbm_bytes"] / 8221440.0
        lines += ["Step traffic (sum of the kernels of one step) = %.0f bytes = %.2fx the algorithmic 8 221 440 bytes (SURVEY.md 8d)."
                  % (summary["step_hbm_bytes"], summary["step_traffic_over_algorithmic"]), ""]
        # the sources and the commit this collection belongs to: bench.py::committed_traffic refuses a collection taken at other sources
        try:
            sys.path.insert(0, ROOT)
            import bench
            summary["csrc_sha16"] = bench.csrc_sha16()
            import subprocess
            summary["commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        except Exception as e:
            sys.stderr.write("summarize_profiles: no source fingerprint (%s)\n" % e)
        json.dump(summary, open(os.path.join(prof, "%s_pmc_cfg3.json" % tag), "w"), indent=1)
    open(os.path.join(prof, "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
