"""CPU: header walk of the exported constraint-system container (host/r1cs_file.hpp, written on a Go box by go/export_r1cs):
counts, the wires K leaves out (what the key loaders need as committed_idx), and that damaged streams are refused."""
import ctypes
import struct

import numpy as np
import pytest

import oracle as O
import r1cs_container as RC
from test_dispatcher_cpu import host  # noqa: F401  (the libzkpor_host.so fixture)


def _parse(host, data):
    counts = (ctypes.c_uint64 * 10)()
    com = (ctypes.c_uint32 * 64)()
    err = ctypes.create_string_buffer(256)
    buf = np.frombuffer(data, dtype=np.uint8)
    rc = host.zkh_r1cs_parse(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(data)), counts, com, ctypes.c_size_t(64), err, ctypes.c_size_t(256))
    return rc, list(counts), list(com), err.value.decode()


def test_counts_and_committed_wires(host):
    S = O.Synth(6, 300, n_public=2, seed=9)
    table, mats = S.r1cs()
    data = RC.from_synth(S, commitments=[(17, [5, 9, 11], [1])])
    rc, c, com, err = _parse(host, data)
    assert rc == 0, err
    assert c[:5] == [S.n_cons, S.n_wires, S.n_public, S.n_wires - S.n_public, table.shape[0]]
    assert c[5:8] == [len(m[1]) for m in mats] and c[8] == 1 and c[9] == 4
    assert com[:4] == [5, 9, 11, 17]          # PrivateCommitted ++ CommitmentIndex: the wires pk.G1.K leaves out besides the public ones


def test_damaged_streams_are_refused(host):
    S = O.Synth(4, 40, n_public=2, seed=2)
    good = RC.from_synth(S)
    assert _parse(host, good)[0] == 0
    cases = {
        "bad magic": b"ZKPR1CS\x02" + good[8:],
        "truncated": good[:-8],
        "trailing bytes": good + b"\0" * 8,
        "truncated header": good[:40],
    }
    bad = bytearray(good); struct.pack_into("<Q", bad, 8 + 8, 0)            # n_wires = 0
    cases["bad wire counts"] = bytes(bad)
    bad = bytearray(good); struct.pack_into("<Q", bad, 8 + 40, 10**12)       # nnzL beyond the stream
    cases["row pointers"] = bytes(bad)
    bad = bytearray(good); struct.pack_into("<Q", bad, 8 + 64, 3)            # commitments that are not there: header eats the table
    cases["commitment"] = bytes(bad)
    for name, data in cases.items():
        rc, _, _, err = _parse(host, data)
        assert rc == 1 and err.startswith("r1cs file:"), name


def test_random_damage_never_escapes_the_header_walk(host):
    """the container is mapped from disk: random byte damage, truncation and growth must end in a refusal (or, when only payload
    bytes changed, in the same counts) — the walk forms no pointer past the stream"""
    import random
    S = O.Synth(5, 120, n_public=2, seed=8)
    good = RC.from_synth(S, commitments=[(9, [3, 4, 6], [1])])
    rc0, c0, _, _ = _parse(host, good)
    assert rc0 == 0
    rng = random.Random(4)
    refused = same = 0
    for _ in range(1200):
        b = bytearray(good)
        mode = rng.randrange(3)
        if mode == 0:
            for _ in range(rng.choice((1, 2, 8))):
                b[rng.randrange(min(len(b), 200))] = rng.randrange(256)      # the header and the commitment info
        elif mode == 1:
            del b[rng.randrange(len(b)):]
        else:
            b += bytes(rng.randrange(256) for _ in range(rng.choice((1, 8, 64))))
        rc, c, _, err = _parse(host, bytes(b))
        if rc == 0:
            same += 1
            assert c[1] < (1 << 32) and c[4] < (1 << 32)
        else:
            refused += 1
            assert err.startswith("r1cs file:")
    assert refused > 600
