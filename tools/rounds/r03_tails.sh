# small partial-sum levels with short chunks (msm_tail_chunk) and the lane-pair scan for the small G2 reduction levels (msm_reduce_scan 1 vs 2)
set -u
OUT=gpurun_out/r03x
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 500 python -m pytest tests/test_msm_gpu.py -x -q -m gpu > $OUT/pytest_msm.txt 2>&1; tail -3 $OUT/pytest_msm.txt
B="python bench.py --steps 6 --warmup 2 --timed-only"
run() { name=$1; shift; timeout 200 $B "$@" 2>$OUT/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['phases_ms_per_proof'])" 2>&1 | tee -a $OUT/tails.txt; }
run old --tail-chunk 0 --reduce-scan 2
run new_default
run tail16 --tail-chunk 16
run tail4 --tail-chunk 4
run tail8_scan2 --reduce-scan 2
run old_again --tail-chunk 0 --reduce-scan 2
