#!/bin/bash
# round 6: "solver_pre_join" — the challenge sponge's inputs evaluated side by side in front of the serial kernel.  Parity (circuit, solver, bench suites), then one worker / two workers A/B
O=gpurun_out/r06ah
mkdir -p $O
timeout 900 python -m pytest tests/test_circuit_gpu.py tests/test_solver_gpu.py tests/test_bench_gpu.py tests/test_prove_batch_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for v in 1 0 1 0; do
  timeout 900 python3 -X faulthandler bench.py --timed-only --steps 10 --warmup 3 --e2e-steps 5 --param solver_pre_join=$v --e2e-sweep "1:0" > $O/bench_$v.json 2> $O/bench_$v.err; rc=$?
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$v.json")); e=d["end_to_end"]
    print("pre_join=$v rc=$rc ms_per_step",round(d["ms_per_step"],1),"dev",e.get("device_phases_ms_per_proof"))
    for r in e.get("sweep", []): print("   one worker", r.get("spec"), r.get("ms_per_proof"), r.get("device_phases_ms_per_proof"), r.get("same_wires"), r.get("note"))
except Exception as ex:
    print("pre_join=$v rc=$rc no line", ex)
PY
  grep "Exception\|rror" $O/bench_$v.err | tail -3 | cut -c1-200
  cp $O/bench_$v.json $O/bench_${v}_$(date +%s).json
done
