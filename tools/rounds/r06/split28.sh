#!/bin/bash
# round 6: BASELINE.json configs[4] — one 2^28 proof, 8 ranks, run one after the other on ONE GPU (tools/split_one_gpu.py); 2^22 first as a rehearsal
O=gpurun_out/r06v
mkdir -p $O
free -g | head -3; cat /sys/fs/cgroup/memory.max 2>/dev/null; nproc
timeout 600 python tools/split_one_gpu.py --log2 22 --wlog 3 --check-h yes --out $O/split_2p22.json > /dev/null 2> $O/split_2p22.err; rc=$?; echo "2^22 rc=$rc"; tail -3 $O/split_2p22.err
if [ $rc -eq 0 ]; then
  timeout 1500 python tools/split_one_gpu.py --log2 28 --wlog 3 --out $O/split_2p28.json > /dev/null 2> $O/split_2p28.err; echo "2^28 rc=$?"; tail -14 $O/split_2p28.err
fi
