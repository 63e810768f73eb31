# round 4: bench.py in circuit mode (small shape first, then the default = zkpor50_1380 compiled circuit), the new circuit tests, a kernel profile of the e2e solve
set -u
OUT=gpurun_out/r04b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py --circuit 5,20,6 --steps 3 --warmup 1 --cpu-log2 14 > $OUT/bench_small_circuit.json 2> $OUT/bench_small_circuit.err; echo "small rc=$?"
tail -3 $OUT/bench_small_circuit.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04b/bench_small_circuit.json"))
print({k:d[k] for k in ("value","ms_per_step","value_uniform")}, d["config"]["scalar_mix_measured"], d["checked"]["ok"], d["checked"]["proofs"])
e=d["end_to_end"]; print({k:e[k] for k in ("value","ms_per_proof","phases_ms_per_proof","checked","same_wires_as_headline")})
PY
timeout 900 python -m pytest tests/test_circuit_gpu.py -x -q > $OUT/pytest_circuit.txt 2>&1; tail -5 $OUT/pytest_circuit.txt
( time timeout 1200 python bench.py --steps 6 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "default rc=$?"; tail -3 $OUT/bench_default.err; cat $OUT/bench_default.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04b/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","value_uniform")}, d["config"]["scalar_mix_measured"], d["checked"])
e=d["end_to_end"]; print({k:e[k] for k in ("value","ms_per_proof","phases_ms_per_proof","device_phases_ms_per_proof","checked","same_wires_as_headline","setup_seconds")})
print(d["phases_ms_per_proof"], d.get("two_in_flight"), d["roofline"]["avg_launch_ms"])
PY
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o e2e -- python tools/rounds/r04/e2e_first.py 50 500 1380 1 > $OUT/e2e_prof.log 2>&1
python tools/rocpd_summary.py $OUT/prof/e2e_results.db $OUT/kernel_stats_e2e.txt > /dev/null 2>&1
rm -rf $OUT/prof/*.db
head -40 $OUT/kernel_stats_e2e.txt
