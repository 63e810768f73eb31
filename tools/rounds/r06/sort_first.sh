#!/bin/bash
# round 6: first contact of the own digit-stream sort (csrc/sort.hip): its parity tests, the MSM / Groth16 suites on top of it, then the timed region
O=gpurun_out/r06c
mkdir -p $O
timeout 600 python -m pytest tests/test_sort_gpu.py -x -q -m gpu > $O/test_sort.log 2>&1; echo "sort rc=$?"; tail -15 $O/test_sort.log
timeout 900 python -m pytest tests/test_msm_gpu.py tests/test_groth16_gpu.py -x -q -m gpu > $O/test_msm.log 2>&1; echo "msm rc=$?"; tail -5 $O/test_msm.log
timeout 900 python3 -X faulthandler bench.py --timed-only --steps 8 --warmup 2 --e2e-steps 6 --e2e-sweep "1:0" > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -5 $O/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06c/bench.json"))
print("headline", d["ms_per_step"], d["end_to_end"]["tail_reserve_cus"], d["end_to_end"].get("device_phases_ms_per_proof"))
print("phases", d.get("phases_ms_per_proof"))
print("tail", d.get("prove_tail_ms_per_proof"), d.get("prove_tail"))
for r in d["end_to_end"].get("sweep", []): print(r.get("spec"), r.get("ms_per_proof"), r.get("k_acc_level1_g1_avg_ms"), r.get("device_phases_ms_per_proof"), r.get("same_wires"), r.get("note"))
print(d.get("checked"))
PY
