#!/bin/bash
# The triplet step at the shipped data shape (50 regions + 35 tokens: R = 51, T = 38) next to BASELINE's 34 x 50, and the kernel
# statistics of the B = 256 step -> gpurun_out/shipped_shape.txt (copy to profiles/<tag>_shipped_shape.txt)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/shipped_shape.txt
mkdir -p "$R/gpurun_out"
(python3 "$R/tools/step_shape.py" 256 34 50 768; python3 "$R/tools/step_shape.py" 256 51 38 768
 python3 "$R/tools/step_shape.py" 256 51 38 768 ragged; python3 "$R/tools/step_shape.py" 32 51 38 768
 python3 "$R/tools/step_shape.py" 64 51 38 768 ragged) 2>&1 | grep "^B " > "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/sshape"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/sshape" -- python3 "$R/tools/step_shape.py" 256 51 38 768 > /dev/null 2>&1
cd "$R"
python3 - >> "$OUT" <<'PY'
import csv, glob, os
f = max(glob.glob('gpurun_out/sshape/*/*kernel_stats.csv'), key=os.path.getmtime)
print('rocprofv3 --kernel-trace --stats of: tools/step_shape.py 256 51 38 768')
for r in list(csv.DictReader(open(f)))[:7]:
    print('  %-100s %5s %12s ns' % (r['Name'][:100], r['Calls'], r['AverageNs']))
PY
cat "$OUT"
