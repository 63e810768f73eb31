#!/bin/bash
# round 6: msm_chain with the second region as an arena of its own: parity files, the full-size headline test (it ran out of memory with two worst-case regions), the bench
O=gpurun_out/r06t
mkdir -p $O
timeout 1500 python -m pytest tests/test_split_gpu.py tests/test_groth16_gpu.py tests/test_msm_gpu.py tests/test_prove_batch_gpu.py tests/test_headline_fullsize_gpu.py tests/test_fullsize_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
timeout 900 python3 -X faulthandler bench.py --timed-only --steps 8 --warmup 2 --e2e-steps 4 > $O/bench_chain1.json 2> $O/bench_chain1.err; echo "rc=$?"; tail -2 $O/bench_chain1.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06t/bench_chain1.json"))
e=d["end_to_end"]
print("ms_per_step",d["ms_per_step"],"tail",d.get("prove_tail_ms_per_proof"),"dev",e.get("device_phases_ms_per_proof"),"failing",e.get("constraints_failing_on_device"))
print("one at a time", e.get("one_proof_at_a_time"))
PY
