#!/bin/bash
# round 6: "ntt_h" — computeH in six transforms (c's coefficients subtracted behind h's inverse coset transform) against gnark's seven.
# Parity first (NTT suite with the new cases, the Groth16 / split suites that run computeH inside proofs), then the headline region + one worker, separate processes
O=gpurun_out/r06ae
mkdir -p $O
timeout 1200 python -m pytest tests/test_ntt_gpu.py tests/test_groth16_gpu.py tests/test_split_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for h in 1 0 1 0; do
  timeout 900 python3 -X faulthandler bench.py --timed-only --steps 10 --warmup 3 --e2e-steps 5 --param ntt_h=$h --e2e-sweep "1:0" > $O/bench_$h.json 2> $O/bench_$h.err; rc=$?
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$h.json")); e=d["end_to_end"]
    print("ntt_h=$h rc=$rc ms_per_step",round(d["ms_per_step"],1),"dev",e.get("device_phases_ms_per_proof"))
    for r in e.get("sweep", []): print("   one worker", r.get("spec"), r.get("ms_per_proof"), r.get("device_phases_ms_per_proof"), r.get("same_wires"), r.get("note"))
except Exception as ex:
    print("ntt_h=$h rc=$rc no line", ex)
PY
  grep "Exception\|rror" $O/bench_$h.err | tail -3 | cut -c1-200
  cp $O/bench_$h.json $O/bench_${h}_$(date +%s).json
done
