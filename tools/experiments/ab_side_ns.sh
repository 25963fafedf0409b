#!/bin/bash
# A/B of the side GEMM's LDS ring depth (diagnostic library, ALADIN_SIDE_NS = 3 / 2) over the shapes that use it:
# rocprofv3 averages of the side GEMM and the step -> gpurun_out/ab_side_ns.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/ab_side_ns.txt
mkdir -p "$R/gpurun_out"; : > "$OUT"
export ALADIN_LIB=$R/aladin_amd/lib/libaladin_hip_diag.so
cd /tmp && export TMPDIR=/tmp
for SHAPE in "256 34 50 768" "256 36 50 768" "256 40 50 768" "256 51 50 768" "256 34 20 768" "64 34 50 768" "256 34 67 768"; do
  for V in 3 2; do
    export ALADIN_SIDE_NS=$V
    T=$(echo $SHAPE | tr ' ' '_')
    rm -rf /tmp/ab_ns_${T}_$V
    timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_ns_${T}_$V -- python3 "$R/tools/step_shape.py" $SHAPE > /tmp/ab_ns.log 2>&1
    F=$(ls /tmp/ab_ns_${T}_$V/*/*kernel_stats.csv 2>/dev/null | head -1)
    if [ -n "$F" ]; then
      echo "stages $V: $(grep 'ms per step' /tmp/ab_ns.log)   side GEMM: $(grep side_gemm "$F" | cut -d, -f1,4 | cut -c1-90)" >> "$OUT"
    else
      echo "stages $V shape $SHAPE: no stats" >> "$OUT"
    fi
  done
done
cat "$OUT"
