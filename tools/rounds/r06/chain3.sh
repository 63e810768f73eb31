#!/bin/bash
# round 6: the full-size files in suite order (the headline test is an isolated child beside a session context that has grown), chain + trim tests
O=gpurun_out/r06u
mkdir -p $O
timeout 1800 python -m pytest tests/test_groth16_gpu.py tests/test_split_gpu.py tests/test_fullsize_gpu.py tests/test_headline_fullsize_gpu.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
