#!/bin/bash
# round 6: with a queue per context stream, does the chain now pay on one-worker (unmasked) tails as well?  The driver's command with msm_chain 2 against 1
O=gpurun_out/r06ac
mkdir -p $O
for ch in 2 1; do
  ZKPOR_BENCH_TRACE=1 timeout 900 python3 -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 --param msm_chain=$ch > $O/bench_chain$ch.json 2> $O/bench_chain$ch.err; rc=$?
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_chain$ch.json")); e=d["end_to_end"]
    print("chain=$ch rc=$rc ms_per_step",round(d["ms_per_step"],1),"tail",round(d["prove_tail_ms_per_proof"],1),"one",round(e["one_proof_at_a_time"]["ms_per_proof"],1),"up",round(e["with_input_upload"]["ms_per_proof"],1),"two_in_flight",round(d["two_in_flight"]["ms_per_step"],1),"uniform",round(d["uniform"]["ms_per_step"],1),"boundary",round(d["boundary"]["ms_per_proof"],1),round(d["boundary"]["one_caller_ms_per_proof"],1),"checked",d["checked"]["ok"],d["checked"]["proofs"])
    print({k_:(v_.get("ms_per_step"), (v_.get("end_to_end") or {}).get("ms_per_proof"), ((v_.get("end_to_end") or {}).get("two_workers") or {}).get("ms_per_proof")) for k_,v_ in (d.get("configs") or {}).items()})
except Exception as ex:
    print("chain=$ch rc=$rc no line", ex)
PY
  grep "Exception\|rror" $O/bench_chain$ch.err | tail -3 | cut -c1-200
done
