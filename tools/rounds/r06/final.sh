#!/bin/bash
# round 6, the end: the whole GPU suite exactly as the driver runs it, smoke, the driver's bench command, and one proof at a time WITHOUT the cross-proof prefetch (the cold-stream latency)
O=gpurun_out/r06ai
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
ZKPOR_BENCH_TRACE=1 timeout 900 python3 -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?
python - <<PY
import json
try:
    d=json.load(open("$O/bench.json")); e=d["end_to_end"]
    print("bench rc=$rc ms_per_step",round(d["ms_per_step"],1),"tail",round(d["prove_tail_ms_per_proof"],1),"one",round(e["one_proof_at_a_time"]["ms_per_proof"],1),"up",round(e["with_input_upload"]["ms_per_proof"],1),"two_in_flight",round(d["two_in_flight"]["ms_per_step"],1),"boundary",round(d["boundary"]["ms_per_proof"],1),"checked",d["checked"]["ok"],d["checked"]["proofs"],"roofline",d["roofline"]["frac"],d["roofline"]["avg_launch_ms"],d["go_toolchain"])
    print({k_:(v_.get("ms_per_step"), (v_.get("end_to_end") or {}).get("ms_per_proof"), ((v_.get("end_to_end") or {}).get("two_workers") or {}).get("ms_per_proof")) for k_,v_ in (d.get("configs") or {}).items()})
except Exception as ex:
    print("bench rc=$rc no line", ex)
PY
grep "Exception\|rror" $O/bench.err | tail -3 | cut -c1-200
timeout 600 python3 bench.py --timed-only --steps 6 --warmup 2 --e2e-workers 1 --no-prefetch > $O/bench_noprefetch.json 2> $O/bench_noprefetch.err; echo "no-prefetch rc=$?"
python -c "
import json; d=json.load(open('$O/bench_noprefetch.json')); print('one worker, no prefetch: ms_per_step', round(d['ms_per_step'],1), d['end_to_end'].get('device_phases_ms_per_proof'))"
