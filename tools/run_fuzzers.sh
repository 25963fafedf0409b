#!/bin/bash
# all GPU fuzzers for S seconds each (default 60), seeds from the clock -> gpurun_out/fuzz.txt (summary lines).
# Each fuzzer's OWN exit status is recorded (PIPESTATUS[0], not the tail's) and the script fails if any fuzzer did.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
S=${1:-60}
SEED=${2:-$(date +%s)}
OUT=$R/gpurun_out/fuzz.txt
mkdir -p "$R/gpurun_out"; : > "$OUT"
FAILED=0
for F in fuzz_parity fuzz_round2 fuzz_round3 fuzz_round4 fuzz_losses; do
  echo "== $F ($S s, seed $SEED)" >> "$OUT"
  timeout $((S * 3 + 120)) python3 "$R/tests/fuzz/$F.py" $S $SEED 2>&1 | grep -v "amdgpu.ids" | tail -6 >> "$OUT"
  RC=${PIPESTATUS[0]}
  echo "exit $RC" >> "$OUT"
  [ "$RC" -ne 0 ] && FAILED=1
done
cat "$OUT"
exit $FAILED
