set -u
OUT=gpurun_out/r04j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_solver_gpu.py tests/test_circuit_gpu.py -x -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
E2E_ROWS=1 timeout 600 python tools/rounds/r04/e2e_first.py 50 500 1380 1 > $OUT/e2e.log 2>&1; tail -7 $OUT/e2e.log
