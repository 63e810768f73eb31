# level-1 kernels pinned to 2 / 4 waves per SIMD against the default (3): bench.py --timed-only with each library build
mkdir -p gpurun_out/r02e
B="python bench.py --steps 3 --warmup 1 --timed-only"
for V in default w4 w2 default; do
  if [ $V = default ]; then L=""; else L="$GRAFT_REPO_ROOT/tools/bin/variants/libzkpor_$V.so"; fi
  echo "== $V" >> gpurun_out/r02e/occupancy.txt
  ZKPOR_LIB=$L timeout 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases_ms_per_proof'])" >> gpurun_out/r02e/occupancy.txt 2>&1
done
cat gpurun_out/r02e/occupancy.txt
