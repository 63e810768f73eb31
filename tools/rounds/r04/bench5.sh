set -u
OUT=gpurun_out/r04m
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 1200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-boundary --other-config-steps 0 --uniform-steps 0 > $OUT/bench_two.json 2> $OUT/bench_two.err ) 2> $OUT/bench_two.time; echo "rc=$?"; tail -3 $OUT/bench_two.err; tail -3 $OUT/bench_two.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04m/bench_two.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["checked"])
e=d["end_to_end"]; print({k:e[k] for k in ("value","ms_per_proof","phases_ms_per_proof","device_phases_ms_per_proof","checked","same_wires_as_headline","two_in_flight")})
print(d["two_in_flight"])
PY
