import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    os.environ.setdefault("ZKPOR_TESTING", "1")   # enables the library's test-only hooks (zkpor_set_param "debug_ntt_fault")
    # a SIGABRT from ANY thread (the HIP / HSA runtimes abort from their own) leaves its native stack behind — in a FILE: pytest captures fd 2
    # during a test and a dying process takes the capture with it (which is how GPUTEST_r04's abort came to look silent)
    trace_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(trace_dir, exist_ok=True)
        os.environ.setdefault("ZKPOR_ABORT_TRACE", os.path.join(trace_dir, "abort_trace.log"))
    except OSError:
        os.environ.setdefault("ZKPOR_ABORT_TRACE", "1")
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); runs through the C ABI of libzkpor.so")
    config.addinivalue_line("markers", "isolated: the test body runs in a child interpreter (threads / several contexts / subprocess drivers): "
                                       "a native crash is a failed test, not a dead session")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once (hipcc cross-compiles without a GPU)
    lib = os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd", "libzkpor.so")
    drivers = [os.path.join(ROOT, "tests", "hostlib", n) for n in ("tree_driver", "witness_driver", "libhostmath.so", "libsolverlogic.so", "libdispatch_gpu.so")]
    if not os.path.exists(lib) or not all(os.path.exists(d) for d in drivers):
        import __graft_entry__
        __graft_entry__.build()


# ---- order of the GPU suite (one abort at test 65 of 497 erased 430 parity tests behind it) ----
# 1. the oracle-parity files of the hot path (MSM, NTT, Poseidon, tree, Groth16 parity cases, a / b / c, solver, generators, key file),
# 2. the rest of the single-context files, 3. the full-size cases, 4. everything that spawns threads, contexts or processes — last.
_ORDER = ["test_sort_gpu", "test_msm_gpu", "test_ntt_gpu", "test_poseidon_gpu", "test_merkle_tree_gpu", "test_groth16_gpu", "test_r1cs_gpu", "test_solver_gpu",
          "test_witgen_gpu", "test_keyfile_gpu", "test_decompress_gpu", "test_cex_gpu", "test_account_totals_gpu", "test_circuit_gpu",
          "test_witness_host_gpu", "test_split_gpu", "test_fullsize_gpu", "test_headline_fullsize_gpu", "test_prove_batch_gpu", "test_pipeline_gpu", "test_dispatcher_gpu",
          "test_bench_gpu"]


def pytest_collection_modifyitems(config, items):
    if os.environ.get("ZKPOR_SUITE_ORDER") == "plain":   # the alphabetical order of rounds 1-4 (tools/rounds/r05_repro_prefix.sh)
        return
    rank = {name: i for i, name in enumerate(_ORDER)}

    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        tier = 1 if item.get_closest_marker("isolated") else 0        # isolated bodies of a parity file go behind every in-process test
        return (tier, rank.get(mod, len(_ORDER) if mod.endswith("_gpu") else -1))

    items.sort(key=key)   # stable: the order inside a file is kept; CPU files (no rank) stay in front, in their own order


_SESSION_ZK = []   # the session fixture's context, while it lives


def pytest_pyfunc_call(pyfuncitem):
    """`@pytest.mark.isolated`: run this one test in a child pytest and report its verdict.  The child sees ZKPOR_ISOLATED_CHILD=1 and
    runs the body in-process; a SIGABRT / SIGSEGV / GPU fault there is this test's failure, with the tail of the child's output."""
    if not pyfuncitem.get_closest_marker("isolated") or os.environ.get("ZKPOR_ISOLATED_CHILD") == "1" or os.environ.get("ZKPOR_SUITE_ORDER") == "plain":
        return None
    if _SESSION_ZK:      # the child proves at full size on the same GPU: the session context's grown scratch (40-60 GB after the full-size files) must not sit beside it
        try:
            _SESSION_ZK[0].trim()
        except Exception:  # noqa: BLE001 — a context that cannot trim is the child's problem to report, not this hook's
            pass
    env = dict(os.environ, ZKPOR_ISOLATED_CHILD="1")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", pyfuncitem.nodeid]
    try:
        r = subprocess.run(cmd, cwd=str(pyfuncitem.config.rootpath), env=env, capture_output=True, text=True, timeout=900)
    except subprocess.TimeoutExpired as e:
        pytest.fail(f"isolated child timed out after 900 s\n{(e.stdout or '')[-3000:]}\n{(e.stderr or '')[-3000:]}", pytrace=False)
    if r.returncode != 0:
        pytest.fail(f"isolated child exited with {r.returncode}\n--- stdout tail\n{r.stdout[-4000:]}\n--- stderr tail\n{r.stderr[-4000:]}", pytrace=False)
    if " skipped" in r.stdout and " passed" not in r.stdout:
        pytest.skip("skipped in the isolated child: " + r.stdout.strip().splitlines()[-1])
    return True


@pytest.fixture(scope="session")
def zk():
    """one libzkpor context on cuda:0 — fails loudly (no fallback) when the HIP library or the GPU is missing"""
    import zkpor
    ctx = zkpor.Context(0)
    _SESSION_ZK.append(ctx)
    yield ctx
    _SESSION_ZK.clear()
    ctx.close()
