# round-3 profiles of the default bench workload on one MI355X (through gpurun; outputs under gpurun_out/r03p, summaries copied into profiles/).
# PMC counters are collected in their own passes, with --kernel-trace only.
set -u
OUT=gpurun_out/r03p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for CH in 48 64; do
  timeout 300 python bench.py --steps 6 --warmup 2 --timed-only --chunk $CH > $OUT/bench_chunk$CH.json 2> $OUT/bench_chunk$CH.err
  python - $OUT/bench_chunk$CH.json $CH <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("chunk",sys.argv[2],"ms_per_step",round(d["ms_per_step"],2),d["phases_ms_per_proof"])
PY
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o timed -- python bench.py --steps 5 --warmup 2 --timed-only > $OUT/bench_timed_only.json 2> $OUT/bench_timed_only.err
python tools/rocpd_summary.py $OUT/prof/timed_results.db $OUT/kernel_stats_timed_only.txt > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python bench.py --log2 26 --steps 1 --warmup 0 --timed-only > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv $OUT/pmc_traffic.json > /dev/null 2>&1
python tools/pmc_valu_summary.py $OUT/pmc_SQ_INSTS_VALU/pmc_counter_collection.csv $OUT/pmc_valu.json r03 > /dev/null 2>&1
rm -rf $OUT/pmc_FETCH_SIZE/pmc_kernel_trace.csv $OUT/pmc_WRITE_SIZE/pmc_kernel_trace.csv $OUT/pmc_SQ_INSTS_VALU/pmc_kernel_trace.csv $OUT/prof/*.db
head -24 $OUT/kernel_stats_timed_only.txt
head -c 500 $OUT/bench_timed_only.json; echo
python -c "
import json; d=json.load(open('$OUT/pmc_traffic.json')); print({k:round(v['hbm_bytes_per_launch']/1e9,2) for k,v in d['kernels'].items()}, d.get('ntt_hbm_bytes_per_computeH',0)/1e9, d.get('ntt_launches_per_computeH'))
d=json.load(open('$OUT/pmc_valu.json')); print({k:(v['launches'], round(v['frac_of_issue_bound_under_pmc'],3)) for k,v in d['kernels'].items()})"
