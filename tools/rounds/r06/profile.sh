#!/bin/bash
# round 6 (VERDICT r05 "next" item 1b): ONE profile set of the headline as it is now — two workers, the tail on its own hardware queues, no reserve, own sort —
# one workload, one binary: kernel stats + timeline of the timed-only command, then FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU / GRBM_GUI_ACTIVE in their own passes.
set -u
OUT=gpurun_out/r06p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMDT="python bench.py --steps 5 --warmup 2 --timed-only"
CMD1="python bench.py --steps 2 --warmup 0 --timed-only"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o timed -- $CMDT > $OUT/bench_timed_only_under_rocprof.json 2> $OUT/bench_timed_only.err
python tools/rocpd_summary.py $OUT/prof/timed_results.db $OUT/kernel_stats_two_workers.txt > /dev/null 2>&1
python tools/rocpd_timeline.py $OUT/prof/timed_results.db > $OUT/timeline_two_workers.txt 2>/dev/null
rm -rf $OUT/prof/*.db
head -40 $OUT/kernel_stats_two_workers.txt | cut -c1-160
for C in SQ_INSTS_VALU FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- $CMD1 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
  ls -la $OUT/pmc_$C/ | head -5
done
python tools/pmc_valu_summary.py $OUT/pmc_SQ_INSTS_VALU/pmc_counter_collection.csv $OUT/pmc_valu.json r06 "$CMD1" > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_WRITE_SIZE/pmc_counter_collection.csv $OUT/pmc_traffic.json > /dev/null 2>&1
python tools/pmc_clock_summary.py $OUT/pmc_GRBM_GUI_ACTIVE/pmc_counter_collection.csv $OUT/clock.txt r06 "$CMD1"
rm -rf $OUT/pmc_SQ_INSTS_VALU $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_GRBM_GUI_ACTIVE
python -c "
import json
d=json.load(open('$OUT/pmc_valu.json')); print({k:(v['launches'], round(v['valu_wave_insts_total']/1e9,3), round(v['frac_of_issue_bound_under_pmc'],3)) for k,v in d['kernels'].items()})
d=json.load(open('$OUT/pmc_traffic.json')); print({k:round(v['hbm_bytes_per_launch']/1e9,2) for k,v in d['kernels'].items()}, d.get('ntt_hbm_bytes_per_computeH',0)/1e9)
d=json.load(open('$OUT/bench_timed_only_under_rocprof.json')); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
cat $OUT/clock.txt | head -30
