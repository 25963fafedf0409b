#!/bin/bash
# rocprofv3 kernel statistics of the sum-of-violations triplet step at B = 256: DENSE_ONLY=1 (arg-max table) / 0 (per pair)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  OUT=$R/gpurun_out/dense$m
  rm -rf "$OUT"; mkdir -p "$OUT"
  DENSE_ONLY=$m rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/tools/experiments/bench_dense_ds.py" > "$OUT/log.txt" 2>&1
  F=$(ls "$OUT"/*/*kernel_stats.csv 2>/dev/null | tail -1)
  echo "== DENSE_ONLY=$m"; tail -1 "$OUT/log.txt"
  [ -n "$F" ] && python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print('%-100s %6s %10s %6s' % (r['Name'][:100], r['Calls'], r['AverageNs'], r['Percentage']))
PY
done
