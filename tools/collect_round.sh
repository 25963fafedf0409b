#!/bin/bash
# Everything profiles/<tag>_* is made from, in ONE gpurun call (run on the GPU box; materialise afterwards in the build container with
# tools/materialise_round.sh <tag>):
#   rocprofv3 kernel stats + PMC passes of bench.py, the bench.py line (unprofiled, last: see below), retrieval: per-data-set timings, kernel stats, PMC
#   (clean input), phase stamps (diagnostic library, if built), the shipped-shape step with its kernel breakdown
# usage: tools/collect_round.sh <tag>
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
# refuse to collect on a library that was not built from this tree's sources (link-time stamp, tools/srchash.py)
if [ "$(cat aladin_amd/lib/libaladin_hip.so.srchash 2>/dev/null)" != "$(python3 tools/srchash.py)" ]; then
  echo "collect_round: aladin_amd/lib/libaladin_hip.so is not built from this tree's sources (make -C aladin_amd/csrc all diag)"; exit 1
fi
python3 tools/bench_retrieval.py > "$O/bench_retrieval.txt" 2>&1
if [ -f aladin_amd/lib/libaladin_hip_diag.so ]; then
  ALADIN_LIB=$R/aladin_amd/lib/libaladin_hip_diag.so python3 tools/retrieval_stamps.py > "$O/retrieval_phase_stamps.txt" 2>&1
fi
bash tools/collect_shipped_shape.sh > /dev/null 2>&1; cp gpurun_out/shipped_shape.txt "$O/shipped_shape.txt"
bash tools/collect_pmc.sh "$TAG" > "$O/collect_pmc.log" 2>&1
bash tools/collect_eval_pmc.sh "${TAG}_eval" "sigma=8" > "$O/collect_eval_pmc.log" 2>&1
bash tools/collect_retrieval_stats.sh "$TAG" > "$O/collect_retrieval_stats.log" 2>&1
# the bench line LAST, after the PMC summary of THESE sources has been written on this box (the same summary is materialised again in
# the build container from the merged raw files): its roofline.traffic / mfma_busy_frac then come from this collection, not from
# a summary of older sources (which bench.py would refuse as stale)
python3 tools/materialise_profiles.py "$TAG" > "$O/materialise_on_box.log" 2>&1
python3 bench.py > "$O/bench_line_unprofiled.json" 2> "$O/bench.err"
tail -2 "$O/bench_retrieval.txt"; head -c 400 "$O/bench_line_unprofiled.json"; echo
