set -u
OUT=gpurun_out/r04v
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
E2E_ROWS=1 timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/prof -o e2e -- python tools/rounds/r04/e2e_first.py 50 500 1380 1 > $OUT/e2e_prof.log 2>&1
python tools/rocpd_timeline.py $OUT/prof/e2e_results.db $OUT/timeline_rep2.txt --from k_hint_inputs --nth 6 --span 700
python tools/rocpd_summary.py $OUT/prof/e2e_results.db $OUT/kernel_stats_e2e.txt > /dev/null 2>&1
rm -rf $OUT/prof/*.db
grep -v simple_timer $OUT/e2e_prof.log | grep "rep \|failing\|trapdoor"
grep "k_gadget_poseidon_coop\|k_solve_chain" $OUT/timeline_rep2.txt | awk '$2>2' | cut -c1-110
