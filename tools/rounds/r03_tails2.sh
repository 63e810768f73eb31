# validation of the short tail chunks / G2 lane-pair scan: the MSM parity cases that touch them, then the default bench line (checked proofs at 2^26)
set -u
OUT=gpurun_out/r03y
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 240 python -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "tail or reduction or refused or windows_chunks or skew or edge" > $OUT/pytest_msm.txt 2>&1; tail -3 $OUT/pytest_msm.txt
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
