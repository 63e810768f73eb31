# round-4 final bench line (after the long-constraint kernel) on one MI355X
set -u
OUT=gpurun_out/r04x
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python bench.py --steps 8 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "default rc=$?"; tail -2 $OUT/bench_default.err; tail -3 $OUT/bench_default.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04x/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","value_uniform")}, d["checked"]["ok"], d["checked"]["proofs"])
e=d["end_to_end"]; print({k:e[k] for k in ("value","ms_per_proof","phases_ms_per_proof","device_phases_ms_per_proof","checked","same_wires_as_headline","two_in_flight")})
o=d["configs"]["zkpor500_200"]; print({k:o[k] for k in ("value","ms_per_step","end_to_end","checked")})
p=d["poseidon_tree"]; print({k:round(v["accounts_per_s"]) for k,v in p["account_leaves"].items()}, p["cex_commitments"]["kernel_ms"], p["build_ms"])
print(d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["two_in_flight"])
print(d["solver_budget"]["device_executor_measured"]["users_side_by_side"], d["solver_budget"]["device_executor_measured"]["users_chained"])
PY
