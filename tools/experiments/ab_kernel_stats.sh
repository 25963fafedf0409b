#!/bin/bash
# per-kernel durations of bench.py under two library builds on ONE box (rocprofv3 --kernel-trace --stats); prints the top rows of each
# usage: tools/experiments/ab_kernel_stats.sh name=path/to/lib.so [name=path ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  D=/tmp/abks_$name; rm -rf "$D"; mkdir -p "$D"
  ALADIN_LIB=$R/$lib rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 "$R/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-eval --graph --repeats 1 --preroll-s 0.3 > "$D/log.txt" 2>&1
  f=$(find "$D" -name '*kernel_stats.csv' | head -1)
  echo "== $name ($lib)"
  python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print('%-60s calls %6s  avg %9.2f us  min %8.2f  max %8.2f' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
P
done
