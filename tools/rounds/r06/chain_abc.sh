#!/bin/bash
# round 6: msm_chain 1 (tails with their own hardware queues only) / 0 / 2 (every tail, the chain stream with a queue of its own), separate processes, one box, twice
O=gpurun_out/r06z
mkdir -p $O
for rep in a b; do
for ch in 1 0 2; do
  timeout 600 python3 bench.py --timed-only --steps 12 --warmup 3 --e2e-steps 4 --param msm_chain=$ch > $O/bench_${rep}_chain$ch.json 2> $O/bench_${rep}_chain$ch.err; rc=$?
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_${rep}_chain$ch.json")); e=d["end_to_end"]
    print("$rep chain=$ch rc=$rc ms_per_step",round(d["ms_per_step"],1),"tail",d.get("prove_tail_ms_per_proof"),"dev",e.get("device_phases_ms_per_proof"))
except Exception as ex:
    print("$rep chain=$ch rc=$rc no line", ex)
PY
done
done
