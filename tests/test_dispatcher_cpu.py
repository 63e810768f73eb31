"""CPU suite: the in-process dispatcher that replaces the reference's Redis task queue (host/prover_host.hpp).
Mirrors src/prover/prover/prover_test.go:TestMockProver — many fake provers, no SNARK, assert exactly-once."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd")
CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p,
                      ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int))


@pytest.fixture(scope="module")
def host():
    so = os.path.join(PKG, "libzkpor_host.so")
    src = os.path.join(PKG, "host", "host_capi.cpp")
    import glob
    deps = [src] + glob.glob(os.path.join(PKG, "host", "*.hpp"))
    if not os.path.exists(so) or max(os.path.getmtime(f) for f in deps) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.zkh_create.restype = ctypes.c_void_p
    for f in ("zkh_count_status", "zkh_count_proofs", "zkh_prove_calls", "zkh_proof_csv"):
        getattr(lib, f).restype = ctypes.c_long
    return lib


def _fake_prover(seen):
    def cb(gpu, height, wit, wlen, out, cap, plen, assets):
        seen.append((gpu, height))
        payload = b"proof-of-%d-by-%s" % (height, ctypes.string_at(wit, wlen))
        ctypes.memmove(out, payload, len(payload))
        plen[0] = len(payload)
        assets[0] = 50
        return 0
    return CB(cb)


def test_exactly_once_with_8_workers(host):
    seen = []
    cb = _fake_prover(seen)
    d = ctypes.c_void_p(host.zkh_create(8, cb, 30))
    n = 2000
    for h in range(n):
        w = b"w%d" % h
        host.zkh_add_witness(d, ctypes.c_int64(h), w, len(w), 0)
        host.zkh_push_task(d, ctypes.c_int64(h))
    made = (ctypes.c_int * 8)()
    host.zkh_run(d, 0, made, 8)
    assert sum(made) == n and all(m >= 0 for m in made)
    assert host.zkh_count_proofs(d) == n
    assert host.zkh_count_status(d, 2) == n and host.zkh_count_status(d, 0) == 0 and host.zkh_count_status(d, 1) == 0
    assert sorted(h for _, h in seen) == list(range(n))          # every batch proven exactly once
    buf = ctypes.create_string_buffer(512); ln = ctypes.c_size_t()
    assert host.zkh_get_proof(d, ctypes.c_int64(7), buf, 512, ctypes.byref(ln)) == 0
    assert buf.raw[:ln.value] == b"proof-of-7-by-w7"
    host.zkh_destroy(d)


def test_stale_queue_entries_and_duplicate_guard(host):
    seen = []
    cb = _fake_prover(seen)
    d = ctypes.c_void_p(host.zkh_create(3, cb, 30))
    for h in range(10):
        host.zkh_add_witness(d, ctypes.c_int64(h), b"x", 1, 0)
    for h in list(range(10)) + [3, 3, 99]:      # a height pushed three times and one with no witness row
        host.zkh_push_task(d, ctypes.c_int64(h))
    host.zkh_insert_proof(d, ctypes.c_int64(5))  # a proof that already exists (crash after CreateProof)
    made = (ctypes.c_int * 3)()
    host.zkh_run(d, 0, made, 3)
    assert sorted(h for _, h in seen) == list(range(10))     # duplicates in the queue are not proven twice
    assert host.zkh_count_proofs(d) == 10 and sum(made) == 9  # height 5 hit the duplicate-proof guard
    assert host.zkh_count_status(d, 2) == 10
    host.zkh_destroy(d)


def test_rerun_picks_received_then_published(host):
    seen = []
    cb = _fake_prover(seen)
    d = ctypes.c_void_p(host.zkh_create(1, cb, 10))
    # rows left behind by crashed provers: two Received, one Published, nothing in the queue
    host.zkh_add_witness(d, ctypes.c_int64(1), b"a", 1, 1)
    host.zkh_add_witness(d, ctypes.c_int64(2), b"b", 1, 0)
    host.zkh_add_witness(d, ctypes.c_int64(3), b"c", 1, 1)
    made = (ctypes.c_int * 1)()
    host.zkh_run(d, 1, made, 1)
    assert [h for _, h in seen] == [3, 1, 2]    # latest Received first, then Published (prover.go:107-137)
    assert host.zkh_count_status(d, 2) == 3
    host.zkh_destroy(d)


def test_rerun_with_several_workers_proves_every_row_once(host):
    """rerun under the in-process dispatcher: GetLatestBatchWitnessByStatus does not claim a row, so without the in-memory claim
    every worker would prove the same latest height (N times the GPU work) and the losers of the CreateProof race would look
    like failures"""
    seen = []
    cb = _fake_prover(seen)
    d = ctypes.c_void_p(host.zkh_create(4, cb, 10))
    n = 40
    for h in range(n):
        host.zkh_add_witness(d, ctypes.c_int64(h), b"r", 1, 1 if h % 3 else 0)   # a mix of Received and Published leftovers
    made = (ctypes.c_int * 4)()
    host.zkh_run(d, 1, made, 4)
    assert all(m >= 0 for m in made) and sum(made) == n
    assert sorted(h for _, h in seen) == list(range(n)) and host.zkh_prove_calls(d) == n
    assert host.zkh_count_proofs(d) == n and host.zkh_count_status(d, 2) == n
    host.zkh_destroy(d)


def test_queue_mode_quits_on_a_height_without_published_row(host):
    """prover.go:150-154: DbErrNotFound from FetchBatchWitness ends Run ("no published status witness in db, so quit")"""
    seen = []
    cb = _fake_prover(seen)
    d = ctypes.c_void_p(host.zkh_create(1, cb, 10))
    for h in (0, 1):
        host.zkh_add_witness(d, ctypes.c_int64(h), b"q", 1, 0)
    for h in (0, 77, 1):                       # 77 has no row: the single worker stops there, height 1 stays Published
        host.zkh_push_task(d, ctypes.c_int64(h))
    made = (ctypes.c_int * 1)()
    host.zkh_run(d, 0, made, 1)
    assert made[0] == 1 and [h for _, h in seen] == [0] and host.zkh_count_status(d, 0) == 1
    host.zkh_destroy(d)


def test_prove_failure_stops_the_worker(host):
    def cb(gpu, height, wit, wlen, out, cap, plen, assets):
        return 1
    cbo = CB(cb)
    d = ctypes.c_void_p(host.zkh_create(1, cbo, 10))
    host.zkh_add_witness(d, ctypes.c_int64(0), b"a", 1, 0)
    host.zkh_push_task(d, ctypes.c_int64(0))
    made = (ctypes.c_int * 1)()
    host.zkh_run(d, 0, made, 1)
    assert made[0] == -1 and host.zkh_count_proofs(d) == 0 and host.zkh_count_status(d, 1) == 1  # stays Received for -rerun
    host.zkh_destroy(d)


def test_shard_range_partitions(host):
    import bench
    for n, world in [(0, 2), (1, 2), (7, 2), (8, 8), (1380, 8), (5, 3)]:
        got = []
        for r in range(world):
            lo, hi = ctypes.c_int64(), ctypes.c_int64()
            host.zkh_shard_range(ctypes.c_int64(n), r, world, ctypes.byref(lo), ctypes.byref(hi))
            assert list(bench.shard_heights(n, r, world)) == list(range(lo.value, hi.value))
            got += list(range(lo.value, hi.value))
        assert got == list(range(n))


def test_proof_row_csv_is_what_the_reference_verifier_reads(host):
    """prover.go:180-236 + dbtool/main.go:260-289: base64.StdEncoding, json.Marshal([][]byte), encoding/csv quoting — checked
    against Python's independent base64 / json / csv implementations, read back the way src/verifier/main.go:127-141 does"""
    import base64, csv, io, json, random
    rng = random.Random(5)
    for trial, raw_len in enumerate((388, 324, 1, 2, 3)):
        raw = bytes(rng.randrange(256) for _ in range(raw_len))
        before, after, root = (bytes(rng.randrange(256) for _ in range(32)) for _ in range(3))
        commit = bytes(rng.randrange(256) for _ in range(32 if trial else 31))
        out = ctypes.create_string_buffer(4096)
        n = host.zkh_proof_csv(raw, ctypes.c_size_t(len(raw)), before, after, root, commit, ctypes.c_size_t(len(commit)),
                               ctypes.c_uint32(7 * trial), ctypes.c_uint32(7 * trial + 1379), 50, ctypes.c_int64(trial), 1, out, ctypes.c_size_t(4096))
        assert n > 0
        text = out.raw[:n].decode()
        # the same row written by an independent CSV writer
        buf = io.StringIO()
        wr = csv.writer(buf, lineterminator="\n")
        wr.writerow(["batch_number", "proof_info", "cex_asset_list_commitments", "account_tree_roots", "batch_commitment",
                     "min_account_index", "max_account_index", "assets_count"])
        b64 = lambda b: base64.b64encode(b).decode()
        wr.writerow([trial, b64(raw), json.dumps([b64(before), b64(after)], separators=(",", ":")), json.dumps([b64(root)]), b64(commit),
                     7 * trial, 7 * trial + 1379, 50])
        assert text == buf.getvalue()
        # and read back as the verifier does: CSV -> JSON arrays -> base64
        rows = list(csv.DictReader(io.StringIO(text)))
        assert len(rows) == 1
        r = rows[0]
        assert base64.b64decode(r["proof_info"]) == raw and base64.b64decode(r["batch_commitment"]) == commit
        assert [base64.b64decode(x) for x in json.loads(r["cex_asset_list_commitments"])] == [before, after]
        assert [base64.b64decode(x) for x in json.loads(r["account_tree_roots"])] == [root]
        assert int(r["batch_number"]) == trial and int(r["max_account_index"]) == 7 * trial + 1379 and int(r["assets_count"]) == 50
    assert host.zkh_proof_csv(b"x", ctypes.c_size_t(1), bytes(32), bytes(32), bytes(32), bytes(32), ctypes.c_size_t(32), 0, 0, 50, ctypes.c_int64(0), 1,
                              ctypes.create_string_buffer(8), ctypes.c_size_t(8)) == -1


def _pipe(host, ns, ng, depth, n, solve_ms, prove_ms, fail_at=-1):
    stats = (ctypes.c_double * 7)()
    heights = (ctypes.c_int64 * n)()
    rc = host.zkh_pipeline_sim(ns, ng, ctypes.c_size_t(depth), ctypes.c_int64(n), solve_ms, prove_ms, ctypes.c_int64(fail_at), stats, heights)
    keys = ("wall_s", "proofs", "solver_busy_s", "solver_blocked_s", "gpu_busy_s", "gpu_starved_s", "max_queued")
    return rc, dict(zip(keys, stats)), list(heights)


def test_pipeline_overlaps_solver_and_gpu_stages(host):
    """solver || GPU (host/prover_host.hpp Pipeline): with enough solver threads the GPU stage sets the pace, the bounded queue
    holds the solvers back, and every batch is proved exactly once; with too few the GPU starves — the two regimes of the
    Amdahl number bench.py prints"""
    n = 24
    # GPU-bound: 8 solvers x 40 ms feed one 10 ms GPU worker: ~n x 10 ms (not n x 50 ms as back-to-back would take)
    rc, st, hs = _pipe(host, 8, 1, 2, n, 40, 10)
    assert rc == 0 and st["proofs"] == n and sorted(hs) == list(range(n))
    assert st["max_queued"] <= 2 and st["solver_blocked_s"] > 0.1
    assert st["wall_s"] < 0.8 * n * 0.050            # back to back would take n x 50 ms; generous for a loaded CI box
    # solver-bound: one 40 ms solver in front of a 10 ms GPU: the GPU waits ~3/4 of the time
    rc, st, hs = _pipe(host, 1, 1, 2, 12, 40, 10)
    assert rc == 0 and st["proofs"] == 12 and st["gpu_starved_s"] > 1.5 * st["gpu_busy_s"]
    # a failing solve stops the pipeline with its code; what was proved before it is still exactly-once
    rc, st, hs = _pipe(host, 2, 1, 2, 12, 5, 5, fail_at=6)
    assert rc == 7 and st["proofs"] < 12 and len(set(hs[: int(st["proofs"])])) == int(st["proofs"])
