# the device solver executor: parity tests, then its rate on the synthetic gadget circuit (tools/bench_solver.py)
set -u
OUT=gpurun_out/r03z
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 240 python -m pytest tests/test_solver_gpu.py -x -q -m gpu > $OUT/pytest_solver.txt 2>&1; tail -25 $OUT/pytest_solver.txt
