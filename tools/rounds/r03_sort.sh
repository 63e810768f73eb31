set -u
OUT=gpurun_out/r03c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for SB in 0 256 512; do
  timeout 300 python bench.py --steps 6 --warmup 2 --timed-only --sort-block $SB > $OUT/bench_sb$SB.json 2> $OUT/bench_sb$SB.err
  python - $OUT/bench_sb$SB.json $SB <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("sort_block",sys.argv[2],"ms_per_step",round(d["ms_per_step"],2),d["phases_ms_per_proof"],d["overlapped_aux_stream_elapsed_ms_per_proof"])
PY
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace256 -o tl -- python bench.py --steps 2 --warmup 1 --timed-only --sort-block 256 > /dev/null 2> $OUT/trace256.err
python tools/timeline_summary.py $OUT/trace256/tl_kernel_trace.csv
timeout 300 python -m pytest tests/test_msm_gpu.py -m gpu -x -q 2>&1 | tail -3
