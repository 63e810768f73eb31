set -u
OUT=gpurun_out/r03g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for F in "--filter-mode 2 --filter-grid 2048" "--filter-mode 2 --filter-grid 1024" "--filter-mode 0" "--filter-mode 2 --filter-grid 2048 --config zkpor500_200" "--filter-mode 0 --config zkpor500_200"; do
  T=$(echo $F | tr -d ' -')
  timeout 300 python bench.py --steps 6 --warmup 2 --timed-only $F > $OUT/bench_$T.json 2> $OUT/bench_$T.err
  python - "$OUT/bench_$T.json" "$F" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2],"ms_per_step",round(d["ms_per_step"],2),d["phases_ms_per_proof"],d["overlapped_aux_stream_elapsed_ms_per_proof"])
PY
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o tl -- python bench.py --steps 2 --warmup 1 --timed-only --filter-mode 2 --filter-grid 2048 > /dev/null 2> $OUT/trace.err
python tools/timeline_summary.py $OUT/trace/tl_kernel_trace.csv | head -40
