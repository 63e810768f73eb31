# round-4 final evidence on one MI355X: the default bench line, the rocprofv3 kernel statistics of the timed region, one end-to-end proof's timeline
set -u
OUT=gpurun_out/r04z
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python bench.py --steps 8 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "default rc=$?"; tail -2 $OUT/bench_default.err; tail -3 $OUT/bench_default.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04z/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","value_uniform")}, d["checked"]["ok"], d["checked"]["proofs"])
e=d["end_to_end"]; print({k:e[k] for k in ("value","ms_per_proof","phases_ms_per_proof","device_phases_ms_per_proof","checked","same_wires_as_headline","two_in_flight","assertions")})
o=d["configs"]["zkpor500_200"]; print({k:o[k] for k in ("value","ms_per_step","end_to_end","checked","setup_seconds")})
p=d["poseidon_tree"]; print(p["account_leaves"], p["cex_commitments"], p["build_ms"])
print(d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["two_in_flight"])
print(d["solver_budget"]["device_executor_measured"])
PY
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o timed -- python bench.py --steps 5 --warmup 2 --timed-only > $OUT/bench_timed_only.json 2> $OUT/bench_timed_only.err
python tools/rocpd_summary.py $OUT/prof/timed_results.db $OUT/kernel_stats_timed_only.txt > /dev/null 2>&1
rm -rf $OUT/prof/*.db
E2E_ROWS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof2 -o e2e -- python tools/rounds/r04/e2e_first.py 50 500 1380 1 > $OUT/e2e_prof.log 2>&1
python tools/rocpd_summary.py $OUT/prof2/e2e_results.db $OUT/kernel_stats_e2e.txt > /dev/null 2>&1
python tools/rocpd_timeline.py $OUT/prof2/e2e_results.db $OUT/timeline_rep2.txt --from k_hint_inputs --nth 6 --span 700
rm -rf $OUT/prof2/*.db
head -16 $OUT/kernel_stats_timed_only.txt
head -c 300 $OUT/bench_timed_only.json; echo
grep -v simple_timer $OUT/e2e_prof.log | grep "rep \|failing\|trapdoor"
