#!/bin/bash
# round 6: the bench suite + the driver's command once more (the upload region over as many proofs as the headline region)
O=gpurun_out/r06ak
mkdir -p $O
timeout 600 python -m pytest tests/test_bench_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
ZKPOR_BENCH_TRACE=1 timeout 900 python3 -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?
python - <<PY
import json
d=json.load(open("$O/bench.json")); e=d["end_to_end"]
print("bench rc=$rc ms_per_step",round(d["ms_per_step"],1),"tail",round(d["prove_tail_ms_per_proof"],1),"one",round(e["one_proof_at_a_time"]["ms_per_proof"],1),"up",round(e["with_input_upload"]["ms_per_proof"],1),e["with_input_upload"]["steps"],"two_in_flight",round(d["two_in_flight"]["ms_per_step"],1),"boundary",round(d["boundary"]["ms_per_proof"],1),"checked",d["checked"]["ok"],d["checked"]["proofs"],"roofline",d["roofline"]["frac"],d["roofline"]["avg_launch_ms"])
print({k_:(v_.get("ms_per_step"), (v_.get("end_to_end") or {}).get("ms_per_proof"), ((v_.get("end_to_end") or {}).get("two_workers") or {}).get("ms_per_proof")) for k_,v_ in (d.get("configs") or {}).items()})
PY
grep "Exception\|rror" $O/bench.err | tail -3 | cut -c1-200
