#!/bin/bash
# timing-only ablation builds of the dense row-step GEMM: aladin_amd/lib/libdr_<n>.so (n = DR_ABLATE)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
cd "$R/aladin_amd/csrc"
make all >/dev/null
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -fvisibility=hidden -DDR_ABLATE=$n -c align_bwd_dense.hip -o /tmp/dr_$n.o
  OBJS=$(ls ../lib/obj/*.o | grep -v align_bwd_dense.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o ../lib/libdr_$n.so $OBJS /tmp/dr_$n.o
done
