#!/bin/bash
# round 6: "poseidon_defer" — the solver's narrow Poseidon launches park their S-box inputs raw, a wide kernel converts behind them.
# Parity first (both modes against the interpreter), then the headline region + one worker at a time, separate processes: 0 (off), 64 (sponge + CEX chains), 4096 (the Merkle levels too)
O=gpurun_out/r06ad
mkdir -p $O
timeout 900 python -m pytest tests/test_circuit_gpu.py -x -q -m gpu -k "small_batches or parked or rows_written or long_call or 500_asset" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for d in 0 64 4096 0 64; do
  timeout 900 python3 -X faulthandler bench.py --timed-only --steps 10 --warmup 3 --e2e-steps 5 --param poseidon_defer=$d --e2e-sweep "1:0" > $O/bench_$d.json 2> $O/bench_$d.err; rc=$?
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$d.json")); e=d["end_to_end"]
    print("defer=$d rc=$rc ms_per_step",round(d["ms_per_step"],1),"dev",e.get("device_phases_ms_per_proof"))
    for r in e.get("sweep", []): print("   one worker", r.get("spec"), r.get("ms_per_proof"), r.get("device_phases_ms_per_proof"), r.get("same_wires"), r.get("note"))
    print("   phases", {k:round(v,1) for k,v in (d.get("phases_ms_per_proof") or {}).items() if "solve" in k or "pos" in k})
except Exception as ex:
    print("defer=$d rc=$rc no line", ex)
PY
  grep "Exception\|rror" $O/bench_$d.err | tail -3 | cut -c1-200
  cp $O/bench_$d.json $O/bench_${d}_$(date +%s).json
done
