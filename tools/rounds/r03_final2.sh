# final validation of the round-3 code on one MI355X: the GPU tests of everything touched since the last full-suite run, then the
# rocprofv3 kernel statistics of the timed-only bench with the final library (copied into profiles/)
set -u
OUT=gpurun_out/r03f2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 330 python -m pytest tests/test_solver_gpu.py tests/test_r1cs_gpu.py tests/test_groth16_gpu.py tests/test_bench_gpu.py tests/test_prove_batch_gpu.py tests/test_msm_gpu.py -x -q -m gpu --durations=8 > $OUT/pytest_subset.txt 2>&1; tail -14 $OUT/pytest_subset.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof -o timed -- python bench.py --steps 5 --warmup 2 --timed-only > $OUT/bench_timed_only.json 2> $OUT/bench_timed_only.err
python tools/rocpd_summary.py $OUT/prof/timed_results.db $OUT/kernel_stats_timed_only.txt > /dev/null 2>&1
rm -rf $OUT/prof/*.db
head -30 $OUT/kernel_stats_timed_only.txt | cut -c1-200
head -c 400 $OUT/bench_timed_only.json; echo
