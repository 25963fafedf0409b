#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline refers to (run on the GPU box through gpurun):
#   1. --kernel-trace --stats summary of `bench.py`            -> profiles/<tag>_kernel_stats.csv
#   2. PMC passes (one counter group per run, kernel-trace only) -> profiles/<tag>_pmc.json
# usage: tools/collect_pmc.sh <tag> [extra bench.py arguments, quoted]
set -u
TAG=${1:-r01}
EXTRA=${2:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT" "$R/profiles"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-eval --graph --repeats 1 --preroll-s 0.3 $EXTRA > "$OUT/bench_stats.log" 2>&1
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  N=$(echo $P | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/pmc_$N" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-eval --graph --repeats 1 --preroll-s 0.05 $EXTRA > "$OUT/pmc_$N.log" 2>&1
done
cd "$R"
# which kernel sources these counters belong to (bench.py refuses a summary whose stamp differs from the tree's)
python3 -c "import bench; print(bench.csrc_hash())" > "$OUT/csrc_hash.txt"
grep -E "^\{" "$OUT/bench_stats.log" | tail -1 > "$OUT/bench_line_under_rocprof.json"
# gpurun merges only gpurun_out/ back: run tools/materialise_profiles.py $TAG in the build container afterwards
# (-> profiles/${TAG}_kernel_stats.csv, ${TAG}_pmc.json with the stamp, ${TAG}_bench_line_under_rocprof.json)
