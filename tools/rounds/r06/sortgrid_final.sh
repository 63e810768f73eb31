#!/bin/bash
# round 6, last: does the sort's persistent grid want another size now that computeH is shorter?  One process, the two-worker region at several grids (A B C D A order)
O=gpurun_out/r06an
mkdir -p $O
timeout 900 python3 -X faulthandler bench.py --timed-only --steps 6 --warmup 2 --e2e-steps 10 --e2e-sweep "2:0:0:128:4096:0:1:1,2:0:0:96:4096:0:1:1,2:0:0:160:4096:0:1:1,2:0:0:192:4096:0:1:1,2:0:0:64:4096:0:1:1,2:0:0:128:4096:0:1:1" > $O/bench.json 2> $O/bench.err; echo rc=$?
python - <<PY
import json
d=json.load(open("$O/bench.json")); e=d["end_to_end"]
print("headline", round(d["ms_per_step"],1))
for r in e.get("sweep", []): print(r.get("spec"), round(r.get("ms_per_proof") or 0,1), r.get("device_phases_ms_per_proof"), r.get("same_wires"), r.get("note"))
PY
