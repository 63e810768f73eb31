#!/bin/bash
# round 6, first call: the masked-stream regression test, then the compute-unit reserve sweep in ONE process (possible now that masked streams live as long as their context)
O=gpurun_out/r06b
mkdir -p $O
timeout 300 python -m pytest tests/test_groth16_gpu.py -x -q -m gpu -k "tail_reserve_cus or prove_tail_matches" -s > $O/test_reserve.log 2>&1; echo "test rc=$?"; tail -5 $O/test_reserve.log
timeout 900 python3 -X faulthandler bench.py --timed-only --steps 8 --warmup 2 --e2e-steps 6 --e2e-sweep "2:16,2:48,2:64,2:32,1:0,2:32:1" > $O/bench_sweep.json 2> $O/bench_sweep.err; echo "rc=$?"; tail -5 $O/bench_sweep.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06b/bench_sweep.json"))
print("headline", d["ms_per_step"], d["end_to_end"]["tail_reserve_cus"], d["end_to_end"].get("device_phases_ms_per_proof"))
print("phases", d.get("phases_ms_per_proof"))
for r in d["end_to_end"].get("sweep", []): print(r.get("spec"), r.get("ms_per_proof"), r.get("k_acc_level1_g1_avg_ms"), r.get("device_phases_ms_per_proof"), r.get("same_wires"), r.get("note"))
PY
