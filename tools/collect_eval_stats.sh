#!/bin/bash
# rocprofv3 kernel statistics of the evaluation workloads (tools/bench_eval.py: configs[2] matching-head retrieval, the COCO-1k /
# COCO-5k alignment-head grids) -> profiles/<tag>_eval_kernel_stats.csv.  usage: tools/collect_eval_stats.sh <tag>
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${TAG}_eval
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/tools/bench_eval.py" > "$OUT/bench_eval.log" 2>&1
cd "$R"
F=$(ls "$OUT"/stats/*/*kernel_stats.csv 2>/dev/null | tail -1)
[ -n "$F" ] && cp "$F" "$OUT/kernel_stats.csv" && head -12 "$OUT/kernel_stats.csv" | cut -c1-200
