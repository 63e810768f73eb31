#!/bin/bash
# round 6: the sums' chains on a stream of their own (msm_chain 1) against one stream (0): parity first, then separate processes
O=gpurun_out/r06r
mkdir -p $O
timeout 1500 python -m pytest tests/test_groth16_gpu.py tests/test_msm_gpu.py tests/test_split_gpu.py tests/test_prove_batch_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
for ch in 1 0; do
  timeout 900 python3 -X faulthandler bench.py --timed-only --steps 8 --warmup 2 --e2e-steps 4 --param msm_chain=$ch > $O/bench_chain$ch.json 2> $O/bench_chain$ch.err; echo "rc=$?"; tail -2 $O/bench_chain$ch.err | cut -c1-300
done
python - <<'PY'
import json
for o in (1,0):
    try:
        d=json.load(open(f"gpurun_out/r06r/bench_chain{o}.json"))
    except Exception as ex:
        print(o, "no json", ex); continue
    e=d["end_to_end"]
    print("msm_chain",o,"ms_per_step",d["ms_per_step"],"tail",d.get("prove_tail_ms_per_proof"),"one_at_a_time",(e.get("one_proof_at_a_time") or {}).get("ms_per_proof"),"dev",e.get("device_phases_ms_per_proof"),"failing",e.get("constraints_failing_on_device"),"checked",d.get("checked"))
    print("   tail phases", (d.get("prove_tail") or {}).get("phases_ms_per_proof"))
PY
