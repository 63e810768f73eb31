#!/bin/bash
# round 6: does the headline depend on which hardware queue the contexts' ORDINARY streams land in?  The driver's command with every context's own stream on a
# hardware queue of its own (stream_own_queue) against the default, separate processes
O=gpurun_out/r06aa
mkdir -p $O
for v in ownq default ownq; do
  if [ $v = ownq ]; then X="--ctx-stream own_queue --param stream_own_queue=1"; else X=""; fi
  ZKPOR_BENCH_TRACE=1 timeout 900 python3 -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 $X > $O/bench_$v.json 2> $O/bench_$v.err; rc=$?
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$v.json")); e=d["end_to_end"]
    print("$v rc=$rc ms_per_step",round(d["ms_per_step"],1),"tail",round(d["prove_tail_ms_per_proof"],1),"one",round(e["one_proof_at_a_time"]["ms_per_proof"],1),"up",round(e["with_input_upload"]["ms_per_proof"],1),"two_in_flight",round(d["two_in_flight"]["ms_per_step"],1),"dev",e.get("device_phases_ms_per_proof"),"checked",d["checked"]["ok"],d["checked"]["proofs"])
except Exception as ex:
    print("$v rc=$rc no line", ex)
PY
  grep "Exception\|rror" $O/bench_$v.err | tail -3 | cut -c1-200
done
