#!/bin/bash
# here (no GPU), after `gpurun -- 'bash tools/run_round_end.sh <tag>'` has merged gpurun_out/: turn the scratch output into the tracked
# profiles/<tag>_* files (rocprofv3 summaries via tools/summarize_profiles.py, the rest copied under stable names)
set -e
cd "$(dirname "$0")/.."; TAG=${1:-r06}; O=gpurun_out/$TAG; P=profiles/$TAG
python tools/summarize_profiles.py $TAG gpurun_out/profiles_$TAG
cp $O/bench.json ${P}_bench_line.json
cp $O/bench_cfg5.json ${P}_cfg5_bench_line.json
cp $O/cfg5_kernel_stats.csv ${P}_cfg5_kernel_stats.csv
cp $O/pmc_cfg5.json ${P}_pmc_cfg5.json
cp $O/pmc_cfg5_FETCH_SIZE.txt ${P}_pmc_cfg5_fetch.txt
cp $O/pmc_cfg5_WRITE_SIZE.txt ${P}_pmc_cfg5_write.txt
cp $O/batch_sweep.txt ${P}_batch_sweep.txt
grep -h "^T=" $O/shape_times.txt > ${P}_shape_times.txt
grep -h "^T=" $O/shape_times_f64.txt > ${P}_shape_times_f64.txt
cp $O/pmc_standalone.txt ${P}_pmc_standalone_B512.txt
grep -v amdgpu.ids $O/fuzz_400.txt | tail -40 > ${P}_fuzz_400.txt
tail -3 $O/pytest.log > ${P}_pytest_gpu_tail.txt
grep -v amdgpu.ids $O/step_grid.txt > ${P}_step_grid_at_head.txt
grep -v amdgpu.ids $O/eager_step_time.txt > ${P}_eager_step_time.txt
grep -v amdgpu.ids $O/overlap_probe.txt > ${P}_overlap_probe.txt
grep -v amdgpu.ids $O/pmc_issue.txt > ${P}_pmc_issue_B512.txt
grep -v amdgpu.ids $O/memset_in_graph.txt > ${P}_memset_in_graph.txt
# the cfg-5 counters: which commit the (source-fingerprinted) collection was taken at
python - <<PY
import json, subprocess
p = "${P}_pmc_cfg5.json"
d = json.load(open(p))
d["commit"] = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
json.dump(d, open(p, "w"), indent=1)
PY
ls -la profiles/ | grep " ${TAG}_" | wc -l
