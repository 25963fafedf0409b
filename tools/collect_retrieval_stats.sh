#!/bin/bash
# rocprofv3 kernel statistics of tools/bench_retrieval.py (configs[2], screened retrieval over easy .. hard data)
# -> gpurun_out/<tag>_retrieval/kernel_stats.csv.   usage: tools/collect_retrieval_stats.sh <tag>
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${TAG}_retrieval
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/bench_retrieval.py" --profile > "$OUT/bench_retrieval.log" 2>&1
cd "$R"
F=$(ls "$OUT"/stats/*/*kernel_stats.csv 2>/dev/null | tail -1)
[ -n "$F" ] && cp "$F" "$OUT/kernel_stats.csv" && cut -d, -f1-4,6,7 "$OUT/kernel_stats.csv" | cut -c1-160 | head -16
T=$(ls "$OUT"/stats/*/*kernel_trace.csv 2>/dev/null | tail -1)
[ -n "$T" ] && cp "$T" "$OUT/kernel_trace.csv"
