#!/bin/bash
# round 6: how much of a compute unit may the digit-stream sort take?  grid (workgroups) x LDS tile, two workers / 32 reserved, in one process
O=gpurun_out/r06d
mkdir -p $O
ZKPOR_ABORT_TRACE=$O/native_trace.log timeout 1200 python3 -X faulthandler bench.py --timed-only --steps 6 --warmup 2 --e2e-steps 6 --e2e-sweep "2:32:0:256:4096,2:32:0:128:4096,2:32:0:256:2048,2:32:0:512:2048,2:32:0:256:1024,2:32:0:512:1024,2:32:0:1024:1024,2:32:0:64:4096,1:0:0:256:2048,1:0:0:128:4096" > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -5 $O/bench.err | cut -c1-300
[ -s $O/native_trace.log ] && head -60 $O/native_trace.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06d/bench.json"))
print("headline", d["ms_per_step"], d["end_to_end"]["tail_reserve_cus"], d["end_to_end"].get("device_phases_ms_per_proof"))
print("phases", d.get("phases_ms_per_proof"))
for r in d["end_to_end"].get("sweep", []): print(r.get("spec"), r.get("ms_per_proof"), r.get("k_acc_level1_g1_avg_ms"), r.get("device_phases_ms_per_proof"), r.get("same_wires"), r.get("note"))
PY
