#!/bin/bash
# round 6: the whole GPU suite as the driver runs it (minus -x: every failure is wanted), then the split file once more on one stream
O=gpurun_out/r06s
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/suite.log 2>&1; echo "suite rc=$?"; tail -8 $O/suite.log
ZKPOR_PARAMS="msm_chain=0" timeout 600 python -m pytest tests/test_split_gpu.py -q -m gpu > $O/split_chain0.log 2>&1; echo "split chain0 rc=$?"; tail -3 $O/split_chain0.log
