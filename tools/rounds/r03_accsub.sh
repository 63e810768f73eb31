# level-1 staging phase (ACC_SUB) 8 / 16 / 32 entries per thread = 6 (register-bound) / 4 / 2 waves per SIMD for the level-1 kernels
set -u
OUT=gpurun_out/r03r
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 5 --warmup 2 --timed-only"
for V in default sub8 sub32 default; do
  if [ $V = default ]; then L=""; else L="$GRAFT_REPO_ROOT/tools/bin/variants/libzkpor_$V.so"; fi
  ZKPOR_LIB=$L timeout 200 $B 2>$OUT/$V.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', round(d['ms_per_step'],2), d['phases_ms_per_proof'])" 2>&1 | tee -a $OUT/accsub.txt
done
