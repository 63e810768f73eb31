#!/bin/bash
# round 6: the whole GPU suite as the driver runs it (-x included) with per-test durations, then the driver's bench command
O=gpurun_out/r06ag
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --durations=80 > $O/suite.log 2>&1; echo "suite rc=$?"; grep -n "passed\|failed\|error" $O/suite.log | tail -4
ZKPOR_BENCH_TRACE=1 timeout 900 python3 -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?
python - <<PY
import json
try:
    d=json.load(open("$O/bench.json")); e=d["end_to_end"]
    print("bench rc=$rc ms_per_step",round(d["ms_per_step"],1),"tail",round(d["prove_tail_ms_per_proof"],1),"one",round(e["one_proof_at_a_time"]["ms_per_proof"],1),"up",round(e["with_input_upload"]["ms_per_proof"],1),"two_in_flight",round(d["two_in_flight"]["ms_per_step"],1),"boundary",round(d["boundary"]["ms_per_proof"],1),"dev",e.get("device_phases_ms_per_proof"),"checked",d["checked"]["ok"],d["checked"]["proofs"])
    print({k_:(v_.get("ms_per_step"), (v_.get("end_to_end") or {}).get("ms_per_proof"), ((v_.get("end_to_end") or {}).get("two_workers") or {}).get("ms_per_proof")) for k_,v_ in (d.get("configs") or {}).items()})
except Exception as ex:
    print("bench rc=$rc no line", ex)
PY
grep "Exception\|rror" $O/bench.err | tail -3 | cut -c1-200
