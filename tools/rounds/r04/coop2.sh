set -u
OUT=gpurun_out/r04h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_r1cs_gpu.py tests/test_circuit_gpu.py tests/test_poseidon_gpu.py tests/test_cex_gpu.py -x -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
timeout 600 python tools/rounds/r04/coop_probe.py > $OUT/coop_probe.txt 2>&1; tail -30 $OUT/coop_probe.txt
E2E_ROWS=1 timeout 600 python tools/rounds/r04/e2e_first.py 50 500 1380 1 > $OUT/e2e.log 2>&1; tail -7 $OUT/e2e.log
