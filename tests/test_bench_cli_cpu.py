"""CPU: bench.py's --gpus flag cannot lie: started bare with N > 1 it must launch N ranks or fail loudly when fewer
GPUs are visible; under a launcher whose WORLD_SIZE differs from --gpus it must refuse instead of printing a line for another
GPU count.  (On this box there is no GPU: both refusals are observable without one.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=300)


def test_gpus_flag_without_enough_devices_fails_loudly():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("8 GPUs are visible")
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "--gpus 8 requested but only" in r.stderr


def test_gpus_flag_must_agree_with_the_launcher():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "WORLD_SIZE=2" in r.stderr and "refusing" in r.stderr
