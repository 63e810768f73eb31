#!/bin/bash
# round 5, the last GPU minutes: compute units reserved for the other worker's solver — 16 / 64 against the default 32, and the digit streams kept inside the mask
O=gpurun_out/r05g
mkdir -p $O
timeout 280 python3 -X faulthandler bench.py --timed-only --steps 6 --warmup 2 --e2e-steps 6 --e2e-sweep "2:16,2:64,2:32:1,2:48" > $O/bench_sweep.json 2> $O/bench_sweep.err; echo "rc=$?"; tail -5 $O/bench_sweep.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05g/bench_sweep.json"))
print("headline", d["ms_per_step"], d["end_to_end"]["tail_reserve_cus"])
for r in d["end_to_end"].get("sweep", []): print(r.get("spec"), r.get("ms_per_proof"), r.get("k_acc_level1_g1_avg_ms"), r.get("device_phases_ms_per_proof"), r.get("same_wires"), r.get("note"))
PY
