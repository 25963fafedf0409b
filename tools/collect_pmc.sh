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
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys, collections, shutil
out, tag = sys.argv[1], sys.argv[2]
stats = sorted(glob.glob(out + '/stats/*/*kernel_stats.csv'), key=os.path.getmtime)       # the newest run's (file names are pids)
if stats:
    shutil.copy(stats[-1], 'profiles/%s_kernel_stats.csv' % tag)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, d in res.items():
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        # rocprofv3 units: KiB.  gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads
        # (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE is exact.
        d['hbm_bytes_corrected'] = (2.0 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024.0
json.dump({'tag': tag, 'note': 'averages per dispatch; FETCH_SIZE doubled per the gfx950 correction', 'kernels': res},
          open('profiles/%s_pmc.json' % tag, 'w'), indent=1, sort_keys=True)
print('wrote profiles/%s_pmc.json with %d kernels' % (tag, len(res)))
PY
grep -E "^\{" "$OUT/bench_stats.log" | tail -1 > "profiles/${TAG}_bench_line.json"
# gpurun merges only gpurun_out/ back: after the call, re-run the python block above in the build
# container on gpurun_out/$TAG (same code) to materialise profiles/${TAG}_*.
