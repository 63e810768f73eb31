set -u
OUT=gpurun_out/r03i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q --durations=6 > $OUT/pytest_full.txt 2>&1
tail -12 $OUT/pytest_full.txt
for F in "--filter-grid 256" "--filter-grid 384" "--filter-grid 512" "--filter-grid 768" "--copy-inputs"; do
  T=$(echo $F | tr -d ' -')
  timeout 300 python bench.py --steps 6 --warmup 2 --timed-only $F > $OUT/bench_$T.json 2> $OUT/bench_$T.err
  python - "$OUT/bench_$T.json" "$F" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2],"ms_per_step",round(d["ms_per_step"],2),d["phases_ms_per_proof"])
PY
done
