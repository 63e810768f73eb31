#!/bin/bash
# level-1 staging variants: tools/bin/variants/libzkpor_sub<N>.so = the product library with ACC_SUB = N (LDS per level-1 workgroup 2 x 256 x (N+1) x 4 B:
# 8 -> 18 KB (8 workgroups per CU by LDS, 6 waves per SIMD by registers), 16 -> 35 KB (4), 32 -> 68 KB (2)).  ZKPOR_LIB=... python bench.py --timed-only
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/zkmerkle-proof-of-solvency_amd/csrc"
make -C "$SRC" -j16 > /dev/null
mkdir -p "$ROOT/tools/bin/variants"
rm -f "$ROOT"/tools/bin/variants/*.so
for N in "$@"; do
  T=$(mktemp -d)
  for f in msm_g1_hot msm_g2_pair msm_g1_cold; do
    EXTRA=""; [ $f = msm_g1_cold ] && EXTRA="-DZK_MUL_NOINLINE"
    (cd "$SRC" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DZK_ACC_SUB=$N $EXTRA -c $f.hip -o $T/$f.o) &
  done
  wait
  OTHERS=$(ls "$SRC"/build/*.o | grep -v "msm_g1_hot\|msm_g2_pair\|msm_g1_cold")
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/bin/variants/libzkpor_sub$N.so" $OTHERS $T/msm_g1_hot.o $T/msm_g2_pair.o $T/msm_g1_cold.o
  rm -rf $T
done
ls -la "$ROOT/tools/bin/variants"
