set -u
OUT=gpurun_out/r04b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_circuit_gpu.py tests/test_solver_gpu.py tests/test_prove_batch_gpu.py tests/test_bench_gpu.py -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
E2E_ROWS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o e2e -- python tools/rounds/r04/e2e_first.py 50 500 1380 1 > $OUT/e2e_prof.log 2>&1
python tools/rocpd_timeline.py $OUT/prof/e2e_results.db $OUT/timeline_rep2.txt --from k_hint_inputs --nth 6 --span 700
rm -rf $OUT/prof/*.db
grep -v simple_timer $OUT/e2e_prof.log | grep "rep \|failing\|trapdoor\|levels"
grep -v "k_solve_level_batched<1>\|k_solve_long\|rocclr\|rocprim\|k_acc_\|k_reduce" $OUT/timeline_rep2.txt | awk '$1<100' | cut -c1-120 | head -60
