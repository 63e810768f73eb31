# round 3, first GPU call: instruction-class issue costs, the new parity tests (h verified; NTT past 2^18), the default bench line,
# and a kernel + HIP API timeline of a timed-only step
set -u
OUT=gpurun_out/r03a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 120 tools/bin/valu_class_bench > $OUT/valu_class.txt 2>&1
tail -35 $OUT/valu_class.txt
timeout 900 python -m pytest tests/test_ntt_gpu.py tests/test_fullsize_gpu.py tests/test_bench_gpu.py tests/test_prove_batch_gpu.py -m gpu -x -q --durations=8 > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
head -c 3000 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
timeout 300 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $OUT/trace -o tl -- python bench.py --steps 2 --warmup 1 --timed-only > $OUT/bench_trace.json 2> $OUT/bench_trace.err
ls -la $OUT/trace | head
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ_INSTS_VALU[A-Z_0-9]*|SQ_VALU[A-Z_0-9]*|SQ_INST_LEVEL[A-Z_0-9]*|SQ_THREAD_CYCLES_VALU|SQ_ACTIVE_INST[A-Z_0-9]*)\b" | sort -u | tr '\n' ' ' > $OUT/valu_counters.txt
cat $OUT/valu_counters.txt
nproc; free -g | head -2
