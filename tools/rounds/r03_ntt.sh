set -u
OUT=gpurun_out/r03j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_ntt_gpu.py tests/test_groth16_gpu.py -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
for F in "" "--no-ntt-fuse" "--filter-grid 128" "--filter-grid 192"; do
  T=$(echo "x$F" | tr -d ' -')
  timeout 300 python bench.py --steps 6 --warmup 2 --timed-only $F > $OUT/bench_$T.json 2> $OUT/bench_$T.err
  python - "$OUT/bench_$T.json" "default $F" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2],"ms_per_step",round(d["ms_per_step"],2),d["phases_ms_per_proof"])
PY
done
