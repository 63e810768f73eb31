# the default bench line of the final round-3 code (driver flags) + the GPU tests touched since the last full-suite run
set -u
OUT=gpurun_out/r03zz
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 3 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err; tail -c 300 $OUT/bench_driver_flags.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03zz/bench_driver_flags.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("checked", {}).get("per_rank_ok_of_total"))
print(json.dumps(d["solver_budget"].get("host_executor_measured"), indent=1)[:1500])
print(json.dumps(d["solver_budget"].get("device_executor_measured"), indent=1)[:2000])
PY
