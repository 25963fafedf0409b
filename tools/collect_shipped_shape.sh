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
# PMC of the same step (one counter group per pass): bytes past L2, L2 hit rate, MFMA busy of the 48-row-class score kernel
cd /tmp
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  N=$(echo $P | cut -d" " -f1)
  rm -rf "$R/gpurun_out/sshape_pmc_$N"
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$R/gpurun_out/sshape_pmc_$N" -- python3 "$R/tools/step_shape.py" 256 51 38 768 > /dev/null 2>&1
done
cd "$R"
python3 - >> "$OUT" <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/sshape_pmc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
print('PMC averages per dispatch (FETCH_SIZE doubled per the gfx950 note; KiB -> MB):')
for k, d in agg.items():
    if 'r48' not in k and 'bwd_rows' not in k and 'side_gemm' not in k:
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    mb = (2 * m.get('FETCH_SIZE', 0) + m.get('WRITE_SIZE', 0)) * 1024 / 1e6
    hit = m.get('TCC_HIT_sum', 0) / max(1.0, m.get('TCC_HIT_sum', 0) + m.get('TCC_MISS_sum', 0))
    print('  %-70s %7.1f MB past L2, L2 hit %.2f, MFMA busy cycles %.3g, LDS bank conflicts %.3g' % (k[:70], mb, hit, m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), m.get('SQ_LDS_BANK_CONFLICT', 0)))
PY
cat "$OUT"
