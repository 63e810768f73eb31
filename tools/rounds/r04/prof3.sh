set -u
OUT=gpurun_out/r04i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
E2E_ROWS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o e2e -- python tools/rounds/r04/e2e_first.py 50 500 1380 1 > $OUT/e2e_prof.log 2>&1
python tools/rocpd_summary.py $OUT/prof/e2e_results.db $OUT/kernel_stats_e2e.txt > /dev/null 2>&1
python - <<'PY'
import sqlite3
db = sqlite3.connect("gpurun_out/r04i/prof/e2e_results.db")
print([r[1] for r in db.execute("pragma table_info(kernels)")])
PY
python tools/rocpd_timeline.py $OUT/prof/e2e_results.db $OUT/timeline_rep2.txt --from k_hint_inputs --nth 6 --span 700
python tools/rocpd_timeline.py $OUT/prof/e2e_results.db $OUT/timeline_all.txt
rm -rf $OUT/prof/*.db
grep -v simple_timer $OUT/e2e_prof.log | tail -6
wc -l $OUT/timeline_rep2.txt $OUT/timeline_all.txt
