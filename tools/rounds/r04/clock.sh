# round 4, first GPU call: the clock the hot kernels actually run at + an unprofiled reference run of the same command.
set -u
OUT=gpurun_out/r04clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L 2>/dev/null | grep -i -E "GRBM_GUI_ACTIVE|GRBM_COUNT|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES" | head -20 > $OUT/counters_available.txt
timeout 400 python bench.py --log2 26 --steps 3 --warmup 1 --timed-only > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
for C in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python bench.py --log2 26 --steps 1 --warmup 0 --timed-only > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python tools/pmc_clock_summary.py $OUT/pmc_GRBM_GUI_ACTIVE/pmc_counter_collection.csv $OUT/clock.txt r04
python tools/pmc_clock_summary.py $OUT/pmc_SQ_BUSY_CYCLES/pmc_counter_collection.csv $OUT/clock_sq.txt r04
# keep the raw rows of the two level-1 kernels only (the full csv is large)
for C in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do
  head -1 $OUT/pmc_$C/pmc_counter_collection.csv > $OUT/rows_$C.csv
  grep -E "k_acc_level1|k_ntt_top29|k_ntt_pass29" $OUT/pmc_$C/pmc_counter_collection.csv | head -60 >> $OUT/rows_$C.csv
  rm -rf $OUT/pmc_$C
done
head -c 600 $OUT/bench_unprofiled.json; echo
