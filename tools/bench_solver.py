"""The device solver executor's rate alone (bench.py's device_executor_leg without the rest of the bench line): python tools/bench_solver.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"))
import bench
import zkpor

ctx = zkpor.Context(0)
try:
    print(json.dumps(bench.device_executor_leg(ctx), indent=1))
finally:
    ctx.close()
