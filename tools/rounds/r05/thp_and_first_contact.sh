#!/bin/bash
# round 5, call 5: (1) the forced transparent-huge-page collapse under a pageable copy (tools/thp_pin_repro.hip) — the candidate root cause of the r04 abort;
# (2) first contact of the round's new code: the bench line in circuit mode, the new / changed GPU tests
O=gpurun_out/r05e
mkdir -p $O
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag /sys/kernel/mm/transparent_hugepage/khugepaged/scan_sleep_millisecs > $O/thp_settings.txt 2>&1
uname -r >> $O/thp_settings.txt
for spec in "3000 0 4" "3000 1 4" "600 1 64" "3000 1 4"; do
  name=$(echo $spec | tr ' ' '_')
  ( time timeout 120 tools/bin/thp_pin_repro $spec ) > $O/thp_$name.log 2>&1; echo "rc=$?" >> $O/thp_$name.log
  tail -4 $O/thp_$name.log
done
# the same through the library, in one long-lived interpreter whose heap is recycled: runtime pin (copy_threads 0) against bounce buffers (4)
( time timeout 300 python tools/repro_prove_tail.py --iters 40 --copy-threads 0 ) > $O/lib_ct0.log 2>&1; echo "rc=$?" >> $O/lib_ct0.log; tail -3 $O/lib_ct0.log
( time timeout 600 python bench.py --circuit 5,20,6 --steps 3 --warmup 1 --no-cpu-baseline --no-boundary --e2e-steps 3 ) > $O/bench_small.json 2> $O/bench_small.err; echo "bench rc=$?"; tail -5 $O/bench_small.err
( time timeout 900 python3 -m pytest tests/test_solver_gpu.py tests/test_circuit_gpu.py tests/test_groth16_gpu.py tests/test_bench_gpu.py tests/test_r1cs_gpu.py tests/test_keyfile_gpu.py -x -q -m gpu -p no:cacheprovider --durations=12 ) > $O/new_tests.log 2>&1; echo "tests rc=$?"; tail -22 $O/new_tests.log
