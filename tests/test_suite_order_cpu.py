"""CPU: the order and the isolation of the GPU suite (tests/conftest.py) — VERDICT r04 weak #2: one native abort at test 65 of 497 erased the 430
parity tests behind it.  The oracle-parity files of the hot path are collected first, everything that spawns threads / contexts / processes
last and behind the `isolated` marker (its body runs in a child interpreter: a crash there is one failed test)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _collected():
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if "::" in l]


def test_parity_files_first_isolated_bodies_last():
    ids = _collected()
    assert len(ids) >= 497
    files = []
    for i in ids:
        f = i.split("::")[0]
        if not files or files[-1] != f:
            files.append(f)
    head = ["tests/test_sort_gpu.py", "tests/test_msm_gpu.py", "tests/test_ntt_gpu.py", "tests/test_poseidon_gpu.py", "tests/test_merkle_tree_gpu.py", "tests/test_groth16_gpu.py", "tests/test_r1cs_gpu.py",
            "tests/test_solver_gpu.py", "tests/test_witgen_gpu.py", "tests/test_keyfile_gpu.py"]
    assert files[:len(head)] == head
    pos = {i: k for k, i in enumerate(ids)}
    iso = [i for i in ids if any(n in i for n in ("test_two_callers_take_turns_on_the_device", "test_host_pointer_staging_survives", "test_two_workers_of_one_gpu", "test_dispatcher_drives_real_proofs",
                                                   "test_row_to_row", "test_pipeline_demo", "test_two_workers_device_tails", "test_headline_", "test_the_ranks_one_after_the_other"))]
    assert len(iso) >= 10
    first_iso = min(pos[i] for i in iso)
    assert all(pos[i] < first_iso for i in ids if i not in iso)          # nothing in-process runs behind an isolated body
    assert pos[next(i for i in ids if "test_fullsize_gpu" in i)] > pos[next(i for i in ids if "test_keyfile_gpu" in i)]   # full-size cases behind the parity files


def test_the_old_order_is_still_available_for_replays():
    env = dict(os.environ, ZKPOR_SUITE_ORDER="plain")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider"], cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    ids = [l for l in r.stdout.splitlines() if "::" in l]
    assert ids[0].startswith("tests/test_account_totals_gpu.py")


def test_a_native_abort_in_an_isolated_body_is_one_failed_test():
    """the mechanism itself: a body that abort()s from a non-Python thread (what GPUTEST_r04 died of) fails ITS test with the child's output,
    the session goes on to the next test and prints its summary"""
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/probe_isolated_case.py", "-q", "-p", "no:cacheprovider"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, r.stdout[-1500:]
    assert "1 failed, 1 passed" in r.stdout and "isolated child exited with" in r.stdout
    assert "Fatal Python error: Aborted" in r.stdout          # the child's own faulthandler output is part of the failure report
