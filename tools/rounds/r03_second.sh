set -u
OUT=gpurun_out/r03d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_witgen_gpu.py tests/test_poseidon_gpu.py tests/test_bench_gpu.py -m gpu -x -q --durations=5 > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
timeout 300 python tools/tier_switch_probe.py 26 > $OUT/tier_switch.json 2> $OUT/tier_switch.err; cat $OUT/tier_switch.json; tail -2 $OUT/tier_switch.err
timeout 700 python bench.py --steps 10 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in ("value","ms_per_step","value_uniform","configs","checked","witness_gen","poseidon_tree"):
    print(k, json.dumps(d.get(k))[:1500])
print("roofline", json.dumps(d["roofline"])[:900])
PY
tail -3 $OUT/bench_default.err
