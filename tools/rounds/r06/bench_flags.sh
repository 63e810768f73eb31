#!/bin/bash
# round 6: the driver's own bench command on the current binary
O=gpurun_out/r06w
mkdir -p $O
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err ) 2> $O/bench.time; echo "bench rc=$?"; tail -3 $O/bench_driver_flags.err | cut -c1-300; tail -3 $O/bench.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06w/bench_driver_flags.json"))
print({k:d.get(k) for k in ("value","ms_per_step","prove_tail_value","prove_tail_ms_per_proof","end_to_end_with_input_upload_value","vs_baseline")})
print(d["checked"])
e=d["end_to_end"]; print({k:e.get(k) for k in ("value","ms_per_proof","phases_ms_per_proof","device_phases_ms_per_proof","checked","same_wires_as_headline","tail_mode","tail_reserve_cus")})
print("one", e.get("one_proof_at_a_time")); print("up", e.get("with_input_upload"))
print(d["roofline"]); print(d["cpu_baseline"]); print(d.get("phases_ms_per_proof")); print(d.get("configs"))
PY
