#!/bin/bash
# A/B of the 24- / 40-word caption classes (diagnostic library: ALADIN_ALIGN_CLASS40 = 1 / 0 = on / whole 16-word tiles) on
# training steps and on the COCO-1k evaluation grid -> gpurun_out/ab_caption_classes.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/ab_caption_classes.txt
mkdir -p "$R/gpurun_out"; : > "$OUT"
export ALADIN_LIB=$R/aladin_amd/lib/libaladin_hip_diag.so
for V in 1 0; do
  export ALADIN_ALIGN_CLASS40=$V
  echo "== ALADIN_ALIGN_CLASS40=$V" >> "$OUT"
  for SHAPE in "256 51 38 768" "256 34 38 768" "256 34 26 768" "256 51 26 768" "32 51 38 768" "256 60 38 768"; do
    timeout 120 python3 "$R/tools/step_shape.py" $SHAPE 2>&1 | grep "^B " >> "$OUT"
  done
  timeout 200 python3 "$R/tools/profile_eval_grid.py" --no-host-profile 2>&1 | grep "precision\|image lengths" >> "$OUT"
  timeout 200 python3 "$R/tools/profile_eval_grid.py" --no-host-profile --img-range 18 51 2>&1 | grep "precision\|image lengths" >> "$OUT"
done
cat "$OUT"
