# final validation of round 3: the whole -m gpu suite, smoke(), and a timeline of the final code
set -u
OUT=gpurun_out/r03u
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest_full.txt 2>&1
tail -10 $OUT/pytest_full.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o tl -- python bench.py --steps 2 --warmup 1 --timed-only > /dev/null 2> $OUT/trace.err
python tools/timeline_summary.py $OUT/trace/tl_kernel_trace.csv > $OUT/timeline.txt; head -40 $OUT/timeline.txt
rm -rf $OUT/trace
