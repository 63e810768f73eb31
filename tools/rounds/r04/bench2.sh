set -u
OUT=gpurun_out/r04c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_circuit_gpu.py tests/test_bench_gpu.py::test_bench_circuit_mode_end_to_end -x -q > $OUT/pytest_circuit.txt 2>&1; tail -4 $OUT/pytest_circuit.txt
( time timeout 1500 python bench.py --steps 8 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "default rc=$?"; tail -2 $OUT/bench_default.err; tail -3 $OUT/bench_default.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04c/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","value_uniform")}, d["checked"]["ok"], d["checked"]["proofs"])
e=d["end_to_end"]; print({k:e[k] for k in ("value","ms_per_proof","phases_ms_per_proof","device_phases_ms_per_proof","checked","same_wires_as_headline","next_proofs_hash_chains_prefetched")})
print(d.get("two_in_flight"), d["roofline"]["avg_launch_ms"])
o=d["configs"]["zkpor500_200"]; print({k:o[k] for k in ("value","ms_per_step","scalar_mix_measured","circuit","end_to_end","checked","setup_seconds")})
PY
timeout 900 python bench.py --steps 4 --warmup 1 --no-prefetch --no-boundary --no-cpu-baseline --other-config-steps 0 --uniform-steps 0 > $OUT/bench_noprefetch.json 2> $OUT/bench_noprefetch.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04c/bench_noprefetch.json"))
e=d["end_to_end"]; print("no prefetch:", {k:e[k] for k in ("value","ms_per_proof","phases_ms_per_proof","checked")})
PY
