#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) over configs[2]'s kernels (recall.hip: sim_screen_kernel on easy .. hard data,
# packers, ground-truth and re-score kernels; tools/bench_retrieval.py --profile) -> gpurun_out/<tag>/pmc_*; materialise with tools/materialise_profiles.py <tag> -> profiles/<tag>_pmc.json
set -u
TAG=${1:-r04_eval}
ONLY=${2:-sigma=8}       # which data set of tools/bench_retrieval.py (substring of its name): the SURVEY 8(d) input (R@1 75 / 41 %) by default
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  N=$(echo $P | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/pmc_$N" -- python3 "$R/tools/bench_retrieval.py" --profile --only "$ONLY" > "$OUT/pmc_$N.log" 2>&1
  tail -1 "$OUT/pmc_$N.log"
done
cd "$R" && python3 -c "import bench; print(bench.csrc_hash())" > "$OUT/csrc_hash.txt"
