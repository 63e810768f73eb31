# round-2 profiles of the default bench workload on one MI355X (run through gpurun; outputs under gpurun_out/r02d, summaries are
# copied into profiles/ by hand).  PMC counters are collected in their own passes, with --kernel-trace only.
set -u
OUT=gpurun_out/r02l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o timed -- python bench.py --steps 5 --warmup 2 --timed-only > $OUT/bench_timed_only.json 2> $OUT/bench_timed_only.err
python tools/rocpd_summary.py $OUT/prof/timed_results.db $OUT/kernel_stats_timed_only.txt > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python bench.py --log2 26 --steps 1 --warmup 0 --timed-only > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv $OUT/pmc_traffic.json > /dev/null 2>&1
python tools/pmc_valu_summary.py $OUT/pmc_SQ_INSTS_VALU/pmc_counter_collection.csv $OUT/pmc_valu.json r02 > /dev/null 2>&1
# keep the merge small: the raw csv files are large
rm -rf $OUT/pmc_FETCH_SIZE/pmc_kernel_trace.csv $OUT/pmc_WRITE_SIZE/pmc_kernel_trace.csv $OUT/pmc_SQ_INSTS_VALU/pmc_kernel_trace.csv
ls -la $OUT $OUT/prof | head -40
head -12 $OUT/kernel_stats_timed_only.txt
cat $OUT/bench_timed_only.json | head -c 600
