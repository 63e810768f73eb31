#!/bin/bash
# round 6: clean A/B in separate processes — no CU reserve + own tail streams against 32 reserved — and the sort without LDS staging inside the first
O=gpurun_out/r06g
mkdir -p $O
timeout 900 python -m pytest tests/test_sort_gpu.py tests/test_groth16_gpu.py -x -q -m gpu -k "sort or every_stream_setting or tail_reserve or per_array or fixed_base" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
ZKPOR_ABORT_TRACE=$O/native_trace_a.log timeout 1200 python3 -X faulthandler bench.py --timed-only --steps 8 --warmup 2 --e2e-steps 6 --tail-reserve-cus 0 --tail-mode 1 --sort-grid 128 --e2e-sweep "2:0:0:128:4096:0:1:0,2:0:0:256:4096:0:1:0,2:0:0:512:4096:0:1:0,2:0:0:256:4096:0:1:1,2:0:0:128:4096:0:1:1,1:0:0:256:4096:0:0:0,1:0:0:128:4096:0:0:1" > $O/bench_a.json 2> $O/bench_a.err; echo "rc=$?"; tail -3 $O/bench_a.err | cut -c1-300
ZKPOR_ABORT_TRACE=$O/native_trace_b.log timeout 900 python3 -X faulthandler bench.py --timed-only --steps 8 --warmup 2 --e2e-steps 6 --tail-reserve-cus 32 --sort-grid 128 > $O/bench_b.json 2> $O/bench_b.err; echo "rc=$?"; tail -3 $O/bench_b.err | cut -c1-300
cat $O/native_trace_*.log 2>/dev/null | head -40
python - <<'PY'
import json
for tag in ("a","b"):
    d=json.load(open(f"gpurun_out/r06g/bench_{tag}.json"))
    print(tag, "headline", d["ms_per_step"], d["end_to_end"]["tail_reserve_cus"], d["end_to_end"].get("device_phases_ms_per_proof"))
    print("  phases", d.get("phases_ms_per_proof"))
    for r in d["end_to_end"].get("sweep", []): print("  ", r.get("spec"), r.get("ms_per_proof"), r.get("k_acc_level1_g1_avg_ms"), r.get("device_phases_ms_per_proof"), r.get("same_wires"), r.get("note"))
PY
