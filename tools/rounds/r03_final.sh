# final measurements of round 3 with the library defaults (per-array streams, fused computeH passes, chunk 64, input-preserving prove tail)
set -u
OUT=gpurun_out/r03q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o timed -- python bench.py --steps 5 --warmup 2 --timed-only > $OUT/bench_timed_only.json 2> $OUT/bench_timed_only.err
python tools/rocpd_summary.py $OUT/prof/timed_results.db $OUT/kernel_stats_timed_only.txt > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python bench.py --log2 26 --steps 1 --warmup 0 --timed-only > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv $OUT/pmc_traffic.json > /dev/null 2>&1
python tools/pmc_valu_summary.py $OUT/pmc_SQ_INSTS_VALU/pmc_counter_collection.csv $OUT/pmc_valu.json r03 > /dev/null 2>&1
rm -rf $OUT/pmc_FETCH_SIZE/pmc_kernel_trace.csv $OUT/pmc_WRITE_SIZE/pmc_kernel_trace.csv $OUT/pmc_SQ_INSTS_VALU/pmc_kernel_trace.csv $OUT/prof/*.db
cp $OUT/pmc_traffic.json profiles/r03_pmc_traffic.json; cp $OUT/pmc_valu.json profiles/r03_pmc_valu.json
head -16 $OUT/kernel_stats_timed_only.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o tl -- python bench.py --steps 2 --warmup 1 --timed-only > /dev/null 2> $OUT/trace.err
python tools/timeline_summary.py $OUT/trace/tl_kernel_trace.csv > $OUT/timeline.txt; head -8 $OUT/timeline.txt
rm -rf $OUT/trace
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
python - $OUT/bench_driver_flags.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in ("value","ms_per_step","value_uniform","configs","checked","two_in_flight","witness_gen","phases_ms_per_proof"):
    print(k, json.dumps(d.get(k))[:700])
print("roofline", json.dumps(d["roofline"])[:1200])
print("boundary", json.dumps(d.get("boundary"))[:600])
print("solver_budget.host_executor", json.dumps(d["solver_budget"].get("host_executor_measured"))[:500])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
PY
timeout 600 python bench.py --steps 10 --warmup 2 --config zkpor500_200 --no-boundary --no-cpu-baseline > $OUT/bench_zkpor500_200.json 2> $OUT/bench_zkpor500_200.err
python -c "
import json; d=json.load(open('$OUT/bench_zkpor500_200.json')); print('zkpor500_200', d['value'], d['ms_per_step'], d['checked']['ok'], d['checked']['proofs'])"
