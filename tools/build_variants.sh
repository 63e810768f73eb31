#!/bin/bash
# kernel build variants for occupancy experiments: tools/bin/variants/libzkpor_w<N>.so = the product library with the two level-1
# accumulation kernels pinned to N waves per SIMD (-DZK_L1_WAVES=N).  Used through ZKPOR_LIB=... python bench.py --timed-only
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/zkmerkle-proof-of-solvency_amd/csrc"
make -C "$SRC" -j16 > /dev/null
mkdir -p "$ROOT/tools/bin/variants"
for W in "$@"; do
  T=$(mktemp -d)
  for f in msm_g1_hot msm_g2_pair; do
    (cd "$SRC" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DZK_L1_WAVES=$W -c $f.hip -o $T/$f.o) &
  done
  wait
  OTHERS=$(ls "$SRC"/build/*.o | grep -v "msm_g1_hot\|msm_g2_pair")
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/bin/variants/libzkpor_w$W.so" $OTHERS $T/msm_g1_hot.o $T/msm_g2_pair.o
  rm -rf $T
done
ls -la "$ROOT/tools/bin/variants"
