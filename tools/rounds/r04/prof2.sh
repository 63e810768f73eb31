set -u
OUT=gpurun_out/r04e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
E2E_ROWS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o e2e -- python tools/rounds/r04/e2e_first.py 50 500 1380 1 > $OUT/e2e_prof.log 2>&1
python tools/rocpd_summary.py $OUT/prof/e2e_results.db $OUT/kernel_stats_e2e_rows.txt > /dev/null 2>&1
rm -rf $OUT/prof/*.db
tail -8 $OUT/e2e_prof.log
head -30 $OUT/kernel_stats_e2e_rows.txt
