#!/bin/bash
# round 5: the driver's own command, on the round's tree (new order, isolated children)
O=gpurun_out/r05s
mkdir -p $O
( time python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=25 ) > $O/suite.log 2>&1; echo "rc=$?" >> $O/suite.log
tail -40 $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
