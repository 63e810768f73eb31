# round-5 measurements of the DEFAULT bench workload (the compiled zkpor50_1380 circuit end to end, two workers, CU-masked tail) on one MI355X.
# PMC counters in their own passes, with --kernel-trace only.
set -u
OUT=gpurun_out/r05p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD1="python bench.py --steps 1 --warmup 0 --timed-only"
# 1. the default line + the worker / reserve sweep in one process
( time timeout 1500 python bench.py --steps 8 --warmup 2 --e2e-steps 4 --e2e-sweep "1:0,2:0,2:16,2:32:1,2:64" > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "default rc=$?"; tail -3 $OUT/bench_default.err; tail -3 $OUT/bench_default.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05p/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","prove_tail_value","prove_tail_ms_per_proof","end_to_end_with_input_upload_value")}, d["checked"]["ok"], d["checked"]["proofs"])
e=d["end_to_end"]; print({k:e.get(k) for k in ("value","ms_per_proof","phases_ms_per_proof","device_phases_ms_per_proof","checked","same_wires_as_headline","one_proof_at_a_time","with_input_upload")})
for r in e.get("sweep", []): print(r.get("spec"), r.get("ms_per_proof"), r.get("k_acc_level1_g1_avg_ms"), r.get("device_phases_ms_per_proof"), r.get("same_wires"), r.get("note"))
print(d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("solver_seconds"))
PY
# 2. kernel stats of the timed-only command
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o timed -- python bench.py --steps 5 --warmup 2 --timed-only > $OUT/bench_timed_only.json 2> $OUT/bench_timed_only.err
python tools/rocpd_summary.py $OUT/prof/timed_results.db $OUT/kernel_stats_timed_only.txt > /dev/null 2>&1
python tools/rocpd_timeline.py $OUT/prof/timed_results.db > $OUT/timeline_timed_only.txt 2>/dev/null
rm -rf $OUT/prof/*.db
head -22 $OUT/kernel_stats_timed_only.txt
# 3. PMC passes
for C in SQ_INSTS_VALU; do   # FETCH_SIZE / WRITE_SIZE: not re-taken this round (GPU minutes): profiles/r04_pmc_traffic.json, corrected to whole proofs by bench.py
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- $CMD1 > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err
done
python tools/pmc_valu_summary.py $OUT/pmc_SQ_INSTS_VALU/pmc_counter_collection.csv $OUT/pmc_valu.json r05 "$CMD1" > /dev/null 2>&1
rm -rf $OUT/pmc_SQ_INSTS_VALU
python -c "
import json
d=json.load(open('$OUT/pmc_valu.json')); print({k:(v['launches'], round(v['frac_of_issue_bound_under_pmc'],3)) for k,v in d['kernels'].items()})"
# 4. the Poseidon tree path: kernel stats, VALU instructions, clock
PCMD="python tools/bench_poseidon.py 27 262144"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/pprof -o pos -- $PCMD > $OUT/poseidon_bench.json 2> $OUT/poseidon_bench.err
python tools/rocpd_summary.py $OUT/pprof/pos_results.db $OUT/poseidon_kernel_stats.txt > /dev/null 2>&1
rm -rf $OUT/pprof
for C in SQ_INSTS_VALU GRBM_GUI_ACTIVE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/ppmc_$C -o pmc -- $PCMD > $OUT/ppmc_$C.json 2> $OUT/ppmc_$C.err
done
python tools/pmc_valu_summary.py $OUT/ppmc_SQ_INSTS_VALU/pmc_counter_collection.csv $OUT/poseidon_pmc_valu.json r05 "$PCMD" > /dev/null 2>&1
python tools/pmc_clock_summary.py $OUT/ppmc_GRBM_GUI_ACTIVE/pmc_counter_collection.csv $OUT/poseidon_clock.txt r05 "$PCMD" > /dev/null 2>&1
rm -rf $OUT/ppmc_SQ_INSTS_VALU $OUT/ppmc_GRBM_GUI_ACTIVE
head -12 $OUT/poseidon_kernel_stats.txt; cat $OUT/poseidon_clock.txt | head -12; cat $OUT/poseidon_bench.json
python -c "
import json; d=json.load(open('$OUT/poseidon_pmc_valu.json')); print({k:(v['launches'], round(v['frac_of_issue_bound_under_pmc'],3), round(v['time_ms_total_under_pmc'],1)) for k,v in d['kernels'].items()})"
# 5. VERDICT r04 item 8, measured: 24-bit windows (11 digits instead of 12, 2^23 buckets per window) against the bucket reduction's cost — one worker, tail + solve in turn
timeout 500 python bench.py --steps 4 --warmup 1 --timed-only --e2e-workers 1 --window 24 > $OUT/bench_window24.json 2> $OUT/bench_window24.err; echo "window24 rc=$?"; tail -2 $OUT/bench_window24.err
timeout 500 python bench.py --steps 4 --warmup 1 --timed-only --e2e-workers 1 > $OUT/bench_window22_one_worker.json 2> $OUT/bench_window22_one_worker.err
python - <<'PY'
import json
for f in ("bench_window24", "bench_window22_one_worker"):
    try:
        d = json.load(open(f"gpurun_out/r05p/{f}.json"))
        print(f, d["ms_per_step"], d["phases_ms_per_proof"], d["roofline"]["avg_launch_ms"], d["end_to_end"].get("bucket_additions_per_proof"))
    except Exception as e:
        print(f, "failed", e)
PY
