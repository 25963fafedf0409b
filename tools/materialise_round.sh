#!/bin/bash
# gpurun_out/<tag>* (merged back by gpurun after tools/collect_round.sh) -> the committed summaries under profiles/
# usage: tools/materialise_round.sh <tag>
set -eu
TAG=${1:-r05}
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
O=gpurun_out/$TAG
python3 tools/materialise_profiles.py "$TAG"
python3 tools/materialise_profiles.py "${TAG}_eval"
cp "$O/bench_line_unprofiled.json" "profiles/${TAG}_bench_line.json"
grep "^{" "$O/bench_retrieval.txt" > "profiles/${TAG}_bench_retrieval.txt"
[ -f "$O/retrieval_phase_stamps.txt" ] && grep -v "amdgpu.ids" "$O/retrieval_phase_stamps.txt" > "profiles/${TAG}_retrieval_phase_stamps.txt"
cp "$O/shipped_shape.txt" "profiles/${TAG}_shipped_shape.txt"
[ -f "gpurun_out/${TAG}_retrieval/kernel_stats.csv" ] && cp "gpurun_out/${TAG}_retrieval/kernel_stats.csv" "profiles/${TAG}_retrieval_kernel_stats.csv"
ls -la profiles | grep "$TAG"
