"""CPU suite: the N > 1 launch path of bench.py with world_size 2 over gloo (no GPU): batch sharding without a
data-path collective, the barrier-bracketed timed region and the MAX-over-ranks elapsed time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import bench
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
heights = list(bench.shard_heights(11, rank, world))
proved = []
def run():
    for h in heights:
        time.sleep(0.01 * (1 + 3 * rank))     # rank 1 is 4x slower: the MAX must win
        proved.append(h)
dt = bench.timed_region(dist, lambda: None, run)
mine = torch.zeros(11, dtype=torch.int64); mine[proved] = 1
dist.all_reduce(mine)                          # test-only bookkeeping, not part of the data path
own = 0.01 * (1 + 3 * rank) * len(heights)
print(json.dumps({"rank": rank, "dt": dt, "own": own, "cover": mine.tolist(), "n": len(heights)}), flush=True)
dist.destroy_process_group()
'''


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, err = p.communicate(timeout=120)
        assert p.returncode == 0, err
        outs.append(json.loads(o.strip().splitlines()[-1]))
    assert outs[0]["cover"] == [1] * 11 and outs[1]["cover"] == [1] * 11   # every batch exactly once, no overlap
    assert {o["n"] for o in outs} == {5, 6}
    slow = max(o["own"] for o in outs)
    for o in outs:                                                         # both ranks report the MAX elapsed time
        assert o["dt"] >= slow * 0.95 and abs(o["dt"] - outs[0]["dt"]) < 1e-9


SPLIT_WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle")); sys.path.insert(0, os.path.join(%(root)r, "zkmerkle-proof-of-solvency_amd"))
import numpy as np, torch, torch.distributed as dist
import oracle as O, zkpor
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 600
sc = O.fr_random(41, n); pts = O.g1_from_scalars(O.fr_random(42, n))
lo, hi = n * rank // world, n * (rank + 1) // world
# the partial sum of this rank's slice: computed by the ORACLE here (no GPU in this test) and handed over as a Jacobian point
aff = O.g1_msm(pts[lo:hi], sc[lo:hi])
part = np.concatenate([aff, O.fp_from_ints([1])[0]]) if aff.any() else np.zeros(12, np.uint64)
mine = torch.from_numpy(part.view(np.int64).copy())
allp = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(allp, mine)                       # the one collective of the single-proof split (96 B per rank)
total = zkpor.g1_jac_sum(np.stack([t.numpy().view(np.uint64) for t in allp]))   # product host code, no device
ok = bool(np.array_equal(O.g1_jac_to_affine(total)[0], O.g1_msm(pts, sc)))
print(json.dumps({"rank": rank, "ok": ok}), flush=True)
dist.destroy_process_group()
'''


def test_msm_split_allgather_two_ranks_gloo(tmp_path):
    script = tmp_path / "split_worker.py"
    script.write_text(SPLIT_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29614", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        o, err = p.communicate(timeout=180)
        assert p.returncode == 0, err
        assert json.loads(o.strip().splitlines()[-1])["ok"]
