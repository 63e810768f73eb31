set -u
OUT=gpurun_out/r04n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_solver_gpu.py tests/test_circuit_gpu.py tests/test_prove_batch_gpu.py -x -q > $OUT/pytest.txt 2>&1; tail -6 $OUT/pytest.txt
( time timeout 1200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-boundary --other-config-steps 0 --uniform-steps 0 --no-two-in-flight > $OUT/bench_checks.json 2> $OUT/bench_checks.err ) 2> $OUT/bench_checks.time; echo "rc=$?"; tail -3 $OUT/bench_checks.err; tail -3 $OUT/bench_checks.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04n/bench_checks.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["checked"]["ok"], d["checked"]["proofs"])
e=d["end_to_end"]; print({k:e[k] for k in ("value","ms_per_proof","phases_ms_per_proof","device_phases_ms_per_proof","checked","same_wires_as_headline")})
PY
