#!/bin/bash
# round 6: a, b, c with the rows in evaluation order (r1cs_order 1) against natural order, separate processes; the r1cs parity tests first
O=gpurun_out/r06q
mkdir -p $O
timeout 600 python -m pytest tests/test_r1cs_gpu.py tests/test_circuit_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for ord in 1 0; do
  timeout 900 python3 -X faulthandler bench.py --timed-only --steps 8 --warmup 2 --e2e-steps 4 --param r1cs_order=$ord > $O/bench_order$ord.json 2> $O/bench_order$ord.err; echo "rc=$?"; tail -2 $O/bench_order$ord.err | cut -c1-300
done
python - <<'PY'
import json
for o in (1,0):
    d=json.load(open(f"gpurun_out/r06q/bench_order{o}.json"))
    e=d["end_to_end"]
    print("r1cs_order",o,"ms_per_step",d["ms_per_step"],"one_at_a_time",e.get("one_proof_at_a_time",{}).get("ms_per_proof"),"dev",e.get("device_phases_ms_per_proof"),"failing",e.get("constraints_failing_on_device"),"checked",d.get("checked",{}).get("ok"))
PY
