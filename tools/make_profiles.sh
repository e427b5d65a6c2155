#!/bin/bash
# Run on the GPU box (via gpurun): collects the rocprofv3 evidence for profiles/.
#   tools/make_profiles.sh <round-tag>       e.g. r01
# 1) kernel trace + stats of the exact bench command   2) separate PMC passes (FETCH_SIZE / WRITE_SIZE)
# Never combines --pmc with sys/hip/hsa tracing.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- \
    python "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline --no-pmc > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o p -- \
    python "$R/tools/pmc_probe.py" > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o p -- \
    python "$R/tools/pmc_probe.py" > "$OUT/pmc_write.log" 2>&1
