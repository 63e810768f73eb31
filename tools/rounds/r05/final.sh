#!/bin/bash
# round 5, last call: (1) the driver's own bench command on the round's final library (own hardware queues for the long streams, one tail at a time, no NULL-stream
# operation in the solver), with faulthandler; (2) the GPU tests the late library changes touch; (3) one worker, for the line's `one_proof_at_a_time` beside it
O=gpurun_out/r05f
mkdir -p $O
( time timeout 420 python3 -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2> $O/bench_driver.time; echo "driver bench rc=$?"; tail -4 $O/bench_driver.err | cut -c1-300; tail -3 $O/bench_driver.time
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r05f/bench_driver.json"))
    print({k:d.get(k) for k in ("value","ms_per_step","prove_tail_value","prove_tail_ms_per_proof","end_to_end_with_input_upload_value")}, d["checked"]["ok"], d["checked"]["proofs"])
    e=d["end_to_end"]; print({k:e.get(k) for k in ("workers_per_gpu","tail_reserve_cus","phases_ms_per_proof","checked","same_wires_as_headline")})
    print("one", (e.get("one_proof_at_a_time") or {}).get("ms_per_proof"), "upload", (e.get("with_input_upload") or {}).get("ms_per_proof"))
    print(d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["cpu_baseline"].get("value"), d["cpu_baseline"].get("solver_seconds"), d["phases_ms_per_proof"])
except Exception as e:
    print("no line:", e)
PY
( time timeout 240 python3 -m pytest tests/test_bench_gpu.py tests/test_solver_gpu.py tests/test_keyfile_gpu.py -x -q -m gpu -p no:cacheprovider -k "circuit or solver or keyfile or external or hint" ) > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-200
( time timeout 200 python3 bench.py --timed-only --steps 5 --warmup 2 --e2e-workers 1 > $O/bench_one_worker.json 2> $O/bench_one_worker.err ); echo "one worker rc=$?"
python -c "
import json; d=json.load(open('$O/bench_one_worker.json')); print(d['value'], d['ms_per_step'], d['end_to_end']['phases_ms_per_proof'])" 2>&1 | tail -2
