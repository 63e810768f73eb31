set -u
OUT=gpurun_out/r04w
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 170 python -m pytest tests/test_dispatcher_gpu.py tests/test_pipeline_gpu.py tests/test_r1cs_gpu.py tests/test_witgen_gpu.py tests/test_poseidon_gpu.py tests/test_cex_gpu.py -x -q > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
