#!/bin/bash
# round 5, call 4: the same replay with pytest's capture OFF (-s): the HIP / HSA runtimes print why they abort on fd 2, and pytest's fd capture ate it
# (gpurun_out/r05b/plain_2.log looked silent for that reason); the abort tracer writes to its own file.
O=gpurun_out/r05d
mkdir -p $O
export ZKPOR_SUITE_ORDER=plain
export ZKPOR_ABORT_TRACE=$PWD/$O/abort_trace.log
export ZKPOR_DEBUG_ADDR=1
FILES="tests/test_account_totals_gpu.py tests/test_cex_gpu.py tests/test_circuit_gpu.py tests/test_decompress_gpu.py tests/test_dispatcher_gpu.py tests/test_fullsize_gpu.py tests/test_groth16_gpu.py"
for i in 1 2 3; do
  ( time timeout 700 python3 -X faulthandler -m pytest $FILES -x -q -s -m gpu -p no:cacheprovider ) > $O/plain_$i.log 2>&1; rc=$?; echo "rc=$rc" >> $O/plain_$i.log
  if [ $rc -eq 134 ]; then break; fi
done
for f in $O/plain_*.log; do echo "== $f"; tail -n 3 $f | cut -c1-200; done
cat $O/abort_trace.log 2>/dev/null | head -60
# ---- and the round's new code, first contact: the bench line in circuit mode (two workers, CU-masked tail), the new GPU tests
unset ZKPOR_SUITE_ORDER ZKPOR_DEBUG_ADDR
( time timeout 600 python bench.py --circuit 5,20,6 --steps 3 --warmup 1 --no-cpu-baseline --no-boundary --e2e-steps 3 ) > $O/bench_small.json 2> $O/bench_small.err; echo "bench rc=$?"; tail -5 $O/bench_small.err
( time timeout 900 python3 -m pytest tests/test_solver_gpu.py tests/test_circuit_gpu.py tests/test_groth16_gpu.py tests/test_bench_gpu.py -x -q -m gpu -p no:cacheprovider --durations=8 ) > $O/new_tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/new_tests.log
