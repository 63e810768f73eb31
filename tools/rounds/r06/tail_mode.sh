#!/bin/bash
# round 6: who gets the compute units — a CU reserve (round 5), or stream priority for the solver and no reserve; and the digit stream of w under the other worker's tail
O=gpurun_out/r06f
mkdir -p $O
timeout 600 python -m pytest tests/test_groth16_gpu.py -x -q -m gpu -k "every_stream_setting or tail_reserve" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
ZKPOR_ABORT_TRACE=$O/native_trace.log timeout 1500 python3 -X faulthandler bench.py --timed-only --steps 6 --warmup 2 --e2e-steps 6 --sort-grid 128 --e2e-sweep "2:32:0:128:4096:0:0,2:0:0:128:4096:0:1,2:0:0:128:4096:0:2,2:8:0:128:4096:0:2,2:16:0:128:4096:0:2,2:16:0:128:4096:0:0,2:32:0:256:4096:0:0,2:0:0:256:4096:0:2" > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -5 $O/bench.err | cut -c1-300
[ -s $O/native_trace.log ] && head -60 $O/native_trace.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06f/bench.json"))
print("headline", d["ms_per_step"], d["end_to_end"]["tail_reserve_cus"], d["end_to_end"].get("device_phases_ms_per_proof"))
print("phases", d.get("phases_ms_per_proof"))
for r in d["end_to_end"].get("sweep", []): print(r.get("spec"), r.get("ms_per_proof"), r.get("k_acc_level1_g1_avg_ms"), r.get("device_phases_ms_per_proof"), r.get("same_wires"), r.get("note"))
PY
