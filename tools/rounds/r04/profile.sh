# round-4 profiles of the DEFAULT bench workload (circuit mode: the compiled zkpor50_1380 circuit, generated scalars) on one MI355X.
# PMC counters in their own passes, with --kernel-trace only.
set -u
OUT=gpurun_out/r04p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_circuit_gpu.py -x -q > $OUT/pytest_circuit.txt 2>&1; tail -3 $OUT/pytest_circuit.txt
( time timeout 1200 python bench.py --steps 6 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "default rc=$?"; tail -2 $OUT/bench_default.err; tail -3 $OUT/bench_default.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04p/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","value_uniform")}, d["checked"]["ok"], d["checked"]["proofs"])
e=d["end_to_end"]; print({k:e[k] for k in ("value","ms_per_proof","phases_ms_per_proof","device_phases_ms_per_proof","checked","same_wires_as_headline","next_proofs_hash_chains_prefetched")})
print(d.get("two_in_flight"), d["roofline"]["avg_launch_ms"], d["configs"])
PY
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o timed -- python bench.py --steps 5 --warmup 2 --timed-only > $OUT/bench_timed_only.json 2> $OUT/bench_timed_only.err
python tools/rocpd_summary.py $OUT/prof/timed_results.db $OUT/kernel_stats_timed_only.txt > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python bench.py --steps 1 --warmup 0 --timed-only > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv $OUT/pmc_traffic.json > /dev/null 2>&1
python tools/pmc_valu_summary.py $OUT/pmc_SQ_INSTS_VALU/pmc_counter_collection.csv $OUT/pmc_valu.json r04 > /dev/null 2>&1
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_INSTS_VALU $OUT/prof/*.db
head -24 $OUT/kernel_stats_timed_only.txt
head -c 400 $OUT/bench_timed_only.json; echo
python -c "
import json; d=json.load(open('$OUT/pmc_traffic.json')); print({k:round(v['hbm_bytes_per_launch']/1e9,2) for k,v in d['kernels'].items()}, d.get('ntt_hbm_bytes_per_computeH',0)/1e9, d.get('ntt_launches_per_computeH'))
d=json.load(open('$OUT/pmc_valu.json')); print({k:(v['launches'], round(v['frac_of_issue_bound_under_pmc'],3)) for k,v in d['kernels'].items()})"
