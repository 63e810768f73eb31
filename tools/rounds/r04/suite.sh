set -u
OUT=gpurun_out/r04s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=25 > $OUT/pytest_gpu.txt 2>&1 ) 2> $OUT/pytest_gpu.time; tail -32 $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
