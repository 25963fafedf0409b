#!/bin/bash
# LDS-side PMC of the score kernel, baseline vs the B-direct experiment (diag build, ALADIN_SCORE_VARIANT=8):
# SQ_INSTS_LDS_LOAD, SQ_ACTIVE_INST_LDS, SQ_LDS_IDX_ACTIVE, SQ_WAIT_INST_LDS, SQ_LDS_DATA_FIFO_FULL, SQ_VALU_MFMA_BUSY_CYCLES.
# -> gpurun_out/lds_pmc_{base,bdirect}/ ; summarised on stdout.  usage: tools/collect_lds_pmc.sh
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export ALADIN_LIB=$R/aladin_amd/lib/libaladin_hip_diag.so
cd /tmp && export TMPDIR=/tmp
for V in base bdirect; do
  if [ $V = bdirect ]; then export ALADIN_SCORE_VARIANT=8; else unset ALADIN_SCORE_VARIANT; fi
  for P in "SQ_INSTS_LDS_LOAD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    N=$(echo $P | cut -d" " -f1)
    rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$R/gpurun_out/lds_pmc_$V/$N" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-eval --graph --repeats 1 --preroll-s 0.05 > "$R/gpurun_out/lds_pmc_$V/$N.log" 2>&1
  done
done
cd "$R"
python3 - <<'PY'
import csv, glob, collections
for v in ('base', 'bdirect'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/lds_pmc_%s/*/*/*counter_collection.csv' % v):
        for r in csv.DictReader(open(f)):
            if 'align_scores16_tall' in r['Kernel_Name']:
                agg[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        print(v, k, {c: round(sum(x) / len(x)) for c, x in sorted(d.items())})
PY
