#!/bin/bash
# round 6: the sort with a compile-time level 0 + generated NTT twiddles: parity first, then the footprint sweep again
O=gpurun_out/r06e
mkdir -p $O
timeout 900 python -m pytest tests/test_sort_gpu.py tests/test_ntt_gpu.py tests/test_msm_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
ZKPOR_ABORT_TRACE=$O/native_trace.log timeout 1200 python3 -X faulthandler bench.py --timed-only --steps 6 --warmup 2 --e2e-steps 6 --e2e-sweep "2:32:0:128:4096,2:32:0:256:4096,2:32:0:192:4096,2:32:0:128:4096:1,2:32:0:256:4096:1,1:0:0:128:4096,1:0:0:128:4096:1,1:0:0:512:4096" > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -5 $O/bench.err | cut -c1-300
[ -s $O/native_trace.log ] && head -60 $O/native_trace.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06e/bench.json"))
print("headline", d["ms_per_step"], d["end_to_end"]["tail_reserve_cus"], d["end_to_end"].get("device_phases_ms_per_proof"))
print("phases", d.get("phases_ms_per_proof"))
for r in d["end_to_end"].get("sweep", []): print(r.get("spec"), r.get("ms_per_proof"), r.get("k_acc_level1_g1_avg_ms"), r.get("device_phases_ms_per_proof"), r.get("same_wires"), r.get("note"))
PY
