#!/bin/bash
# round 6: the sharded computeH in six transforms (c not exchanged after step 1) — the split suite, then configs[4] again: one 2^28 proof, 8 ranks one after the other on ONE GPU
O=gpurun_out/r06af
mkdir -p $O
timeout 900 python -m pytest tests/test_split_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 600 python tools/split_one_gpu.py --log2 22 --wlog 3 --check-h yes --out $O/split_2p22.json > /dev/null 2> $O/split_2p22.err; rc=$?; echo "2^22 rc=$rc"; tail -3 $O/split_2p22.err
if [ $rc -eq 0 ]; then
  timeout 1500 python tools/split_one_gpu.py --log2 28 --wlog 3 --out $O/split_2p28.json > /dev/null 2> $O/split_2p28.err; echo "2^28 rc=$?"; tail -14 $O/split_2p28.err
fi
