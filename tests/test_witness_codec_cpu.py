"""CPU: the witness row codec (host/witness_codec.hpp: base64(s2(gob(BatchCreateUserWitness))), src/witness/witness/witness.go:215-232,
src/utils/utils.go:704-742) against (1) the worked example of the encoding/gob documentation, byte for byte, (2) hand-written
expected bytes of small values, (3) an independent pure-Python gob / s2 implementation in both directions — including streams laid
out the way another encoder might (other type ids, other definition order, zero fields sent explicitly, repeat-offset copies)."""
import base64
import ctypes

import pytest

import gobs2 as G
from test_dispatcher_cpu import host  # noqa: F401


def _call(host, fn, *args, cap=1 << 26):
    out = ctypes.create_string_buffer(cap)
    getattr(host, fn).restype = ctypes.c_long
    n = getattr(host, fn)(*args, out, ctypes.c_size_t(cap))
    assert n >= 0, n
    return out.raw[:n]


def test_gob_documentation_example_bytes(host):
    """encoding/gob package documentation, "type Point struct {X, Y int}": the type definition message and Point{22, 33}"""
    doc = bytes.fromhex("1f ff 81 03 01 01 05 50 6f 69 6e 74 01 ff 82 00 01 02 01 01 58 01 04 00 01 01 59 01 04 00 00 00"
                        " 07 ff 82 01 2c 01 42 00".replace(" ", ""))
    assert _call(host, "zkh_gob_point_example", cap=256) == doc
    v, types = G.gob_decode(doc)                       # ... and the independent decoder reads the documentation's bytes
    assert v == {"X": 22, "Y": 33} and types[65]["name"] == "Point"


def test_integer_encodings_by_hand():
    # documentation: 7 -> 07, 256 -> FE 01 00; signed: bit 0 says complement, -129 -> (^-129 << 1) | 1 = 257 -> FE 01 01
    assert G.g_uint(7) == b"\x07" and G.g_uint(256) == b"\xfe\x01\x00" and G.g_int(-129) == b"\xfe\x01\x01" and G.g_int(65) == b"\xff\x82"


def _synth(host, seed, users, assets, cex, level=1, stage=2):
    return _call(host, "zkh_witness_synth_encode", ctypes.c_uint64(seed), users, assets, cex, level, stage)


def test_one_user_witness_bytes_by_hand(host):
    """the value message of a 1-user, 1-asset, 0-CEX-asset witness written out by hand from the format rules"""
    g = _synth(host, 5, 1, 1, 0, stage=0)
    full = G.synth_witness(5, 1, 1, 0)
    op = full["CreateUserOps"][0]; a = op["Assets"][0]
    u, b = G.g_uint, G.g_bytes
    assert a["Index"] == 0 and a["Debt"] == 0 and a["Margin"] == 0 and a["Loan"] and a["Equity"]
    asset = u(2) + u(a["Equity"]) + u(2) + u(a["Loan"]) + u(2) + u(a["PortfolioMargin"]) + u(0)   # Index, Debt, Margin are 0: skipped
    body = (u(1) + b(full["BatchCommitment"]) + u(1) + b(full["AccountTreeRoot"]) + u(1) + b(full["BeforeCEXAssetsCommitment"])
            + u(1) + b(full["AfterCEXAssetsCommitment"]) + u(1) + u(full["MinAccountIndex"]) + u(1) + u(full["MaxAccountIndex"])
            + u(2)                                            # field 7 CreateUserOps (6 BeforeCexAssets is empty: skipped)
            + u(1)                                            # one operation
            + u(1) + u(1) + asset                             # field 0 Assets: one element
            + u(1) + u(op["AccountIndex"]) + u(1) + b(op["AccountIdHash"])
            + u(1) + u(28) + b"".join(b(p) for p in op["AccountProof"]) + u(0)
            + u(0))
    value_msg = G.g_int(65) + body
    assert g.endswith(u(len(value_msg)) + value_msg)


@pytest.mark.parametrize("users,assets,cex,level", [(0, 0, 0, 1), (1, 1, 0, 0), (3, 5, 4, 1), (40, 50, 500, 1)])
def test_cpp_encoder_read_by_independent_decoder(host, users, assets, cex, level):
    col = _synth(host, 11 + users, users, assets, cex, level)
    z = base64.b64decode(col, validate=True)
    g = G.s2_decode(z)
    assert g == _synth(host, 11 + users, users, assets, cex, level, stage=0)
    v, _ = G.gob_decode(g)
    full = G.synth_witness(11 + users, users, assets, cex)
    assert G.normalise(v, full) == full
    if users >= 40:                                           # matched copies found even in mostly random synthetic values (compression proper: test_s2_blocks)
        assert len(z) < len(g)


def test_foreign_stream_layout_is_accepted_and_preserved(host):
    """a stream as ANOTHER encoder may lay it out — different type ids, inner types defined after the types that use them, zero
    fields sent explicitly, literal-only s2 — decodes (utils.DecodeBatchWitness), and re-encoding loses nothing"""
    full = G.synth_witness(99, 3, 4, 5)
    B, U, S = 5, 3, 6
    types = {
        70: {"kind": "struct", "name": "BatchCreateUserWitness", "fields": [("BatchCommitment", B), ("AccountTreeRoot", B), ("BeforeCEXAssetsCommitment", B),
                                                                          ("AfterCEXAssetsCommitment", B), ("MinAccountIndex", U), ("MaxAccountIndex", U),
                                                                          ("BeforeCexAssets", 90), ("CreateUserOps", 80)]},
        90: {"kind": "slice", "name": "[]utils.CexAssetInfo", "elem": 91},
        91: {"kind": "struct", "name": "CexAssetInfo", "fields": [("TotalEquity", U), ("TotalDebt", U), ("BasePrice", U), ("Symbol", S), ("Index", U),
                                                                 ("LoanCollateral", U), ("MarginCollateral", U), ("PortfolioMarginCollateral", U),
                                                                 ("LoanRatios", 92), ("MarginRatios", 92), ("PortfolioMarginRatios", 92)]},
        92: {"kind": "array", "name": "[12]utils.TierRatio", "elem": 93, "len": 12},
        93: {"kind": "struct", "name": "TierRatio", "fields": [("BoundaryValue", 94), ("Ratio", U), ("PrecomputedValue", 94)]},
        94: {"kind": "gobenc", "name": "Int"},
        80: {"kind": "slice", "name": "", "elem": 81},
        81: {"kind": "struct", "name": "CreateUserOperation", "fields": [("Assets", 82), ("AccountIndex", U), ("AccountIdHash", B), ("AccountProof", 84)]},
        82: {"kind": "slice", "name": "[]utils.AccountAsset", "elem": 83},
        83: {"kind": "struct", "name": "AccountAsset", "fields": [("Index", U), ("Equity", U), ("Debt", U), ("Loan", U), ("Margin", U), ("PortfolioMargin", U)]},
        84: {"kind": "array", "name": "[28][]uint8", "elem": B, "len": 28},
    }
    def strip_none(v):   # nil *big.Int fields are never sent
        if isinstance(v, dict):
            return {k: strip_none(x) for k, x in v.items() if x is not None}
        if isinstance(v, list):
            return [strip_none(x) for x in v]
        return v
    stream = G.gob_encode(types, [70, 80, 81, 84, 82, 83, 90, 91, 92, 93, 94], 70, strip_none(full), send_zero=True)
    col = base64.b64encode(G.s2_literal_block(stream))
    err = ctypes.create_string_buffer(256)
    back = _call(host, "zkh_witness_reencode", col, ctypes.c_size_t(len(col)), 0, 1, cap=1 << 24)
    v, _ = G.gob_decode(G.s2_decode(base64.b64decode(back)))
    assert G.normalise(v, full) == full
    # with the expansion of utils.DecodeBatchWitness: every user's list becomes the dense 500-entry list, stored entries at their index
    dense = _call(host, "zkh_witness_reencode", col, ctypes.c_size_t(len(col)), 1, 1, cap=1 << 24)
    vd, _ = G.gob_decode(G.s2_decode(base64.b64decode(dense)))
    for op, ref in zip(vd["CreateUserOps"], full["CreateUserOps"]):
        assert len(op["Assets"]) == 500
        for p, a in enumerate(op["Assets"]):
            assert a.get("Index", 0) == p
        for a in ref["Assets"]:
            got = op["Assets"][a["Index"]]
            assert all(got.get(k, 0) == a[k] for k in a)
        assert sum(1 for a in op["Assets"] if len(a) > 1) == len({a["Index"] for a in ref["Assets"]})
    del err


def test_s2_blocks(host):
    dec = lambda b: _call(host, "zkh_s2", b, ctypes.c_size_t(len(b)), 1, 0, ctypes.create_string_buffer(256), ctypes.c_size_t(256), cap=1 << 22) if False else None
    def s2(b, decode, level=1):
        out = ctypes.create_string_buffer(1 << 22); err = ctypes.create_string_buffer(256)
        host.zkh_s2.restype = ctypes.c_long
        n = host.zkh_s2(b, ctypes.c_size_t(len(b)), decode, level, out, ctypes.c_size_t(1 << 22), err, ctypes.c_size_t(256))
        return (out.raw[:n], None) if n >= 0 else (None, err.value.decode())
    # hand-made blocks: literal "abcd" + copy1(offset 4, length 4) ; run-length through an overlapping copy2
    assert s2(bytes([8, 3 << 2]) + b"abcd" + bytes([1 | (0 << 2) | (0 << 5), 4]), 1)[0] == b"abcdabcd"
    assert s2(bytes([10, 0]) + b"x" + bytes([2 | (8 << 2), 1, 0]), 1)[0] == b"x" * 10
    # copy4, and S2's repeat (copy1 with offset 0): lengths 4 + 3 (short form) and the one-byte extension 5 -> next byte + 4 + 4
    assert s2(bytes([9, 4 << 2]) + b"hello" + bytes([3 | (3 << 2), 5, 0, 0, 0]), 1)[0] == b"hellohell"
    blk = bytes([6 + 4 + 7 + 29, 5 << 2]) + b"abcdef" + bytes([1 | (0 << 2), 6]) + bytes([1 | (3 << 2), 0]) + bytes([1 | (5 << 2), 0, 21])
    exp = b"abcdef" + b"abcd" + b"efabcde" + (b"fabcde" * 6)[:29]
    assert G.s2_decode(blk) == exp and s2(blk, 1)[0] == exp
    # damaged blocks are refused
    for bad in (bytes([8, 3 << 2]) + b"abc", bytes([4, 1, 9]), bytes([200, 0]) + b"x", bytes([3, 2 << 2]) + b"abcd"):
        assert s2(bad, 1)[0] is None
    # encoder -> independent decoder, literal-only and matched, on text with repeats and on incompressible bytes
    import random
    rng = random.Random(3)
    samples = [b"", b"a", b"abcabcabcabcabcabcabcabcabcabc" * 40, bytes(rng.randrange(256) for _ in range(5000)),
               (b"0123456789abcdef" * 5000) + bytes(rng.randrange(4) for _ in range(70000))]
    for src in samples:
        for level in (0, 1):
            z, _ = s2(src, 0, level)
            assert G.s2_decode(z) == src and s2(z, 1)[0] == src
        if len(src) > 1000 and src[:4] == b"0123":
            assert len(s2(src, 0, 1)[0]) < len(src) // 3


def test_damaged_rows_are_refused_not_crashed_on(host):
    """WitnessData comes out of a database: thousands of random single- and multi-byte mutations of a valid row (inside the base64,
    the s2 block and the gob stream) must each end in an error or in a decoded witness — never in a crash or a runaway allocation"""
    import random
    col = _synth(host, 7, 3, 4, 6)
    z = base64.b64decode(col)
    g = G.s2_decode(z)
    rng = random.Random(11)
    err = ctypes.create_string_buffer(256)
    out = ctypes.create_string_buffer(1 << 22)
    host.zkh_witness_reencode.restype = ctypes.c_long
    outcomes = {"ok": 0, "error": 0}

    def feed(column):
        n = host.zkh_witness_reencode(column, ctypes.c_size_t(len(column)), 1, 0, out, ctypes.c_size_t(1 << 22), err, ctypes.c_size_t(256))
        outcomes["ok" if n >= 0 else "error"] += 1

    for trial in range(1500):
        layer = trial % 3
        src = bytearray(col if layer == 0 else z if layer == 1 else g)
        for _ in range(rng.choice((1, 1, 2, 5))):
            pos = rng.randrange(len(src))
            mode = rng.randrange(4)
            if mode == 0:
                src[pos] = rng.randrange(256)
            elif mode == 1:
                src[pos] ^= 1 << rng.randrange(8)
            elif mode == 2:
                del src[pos:pos + rng.choice((1, 3, 40))]
            else:
                src[pos:pos] = bytes(rng.randrange(256) for _ in range(rng.choice((1, 9))))
            if not src:
                src = bytearray(b"A")
        if layer == 0:
            feed(bytes(src))
        elif layer == 1:
            feed(base64.b64encode(bytes(src)))
        else:
            feed(base64.b64encode(G.s2_literal_block(bytes(src))))
    # huge counts and lengths in front of nothing
    for evil in (b"\xf8" + b"\xff" * 8, b"\x05\xff\x81" + b"\xf8" + b"\x7f" * 8, b"\x03\xff\x82\x00"):
        feed(base64.b64encode(G.s2_literal_block(evil)))
    feed(base64.b64encode(bytes([0xff, 0xff, 0xff, 0xff, 0x0f, 0x00]) + b"x"))      # s2 block claiming 4 GiB
    assert outcomes["error"] > 500 and outcomes["ok"] + outcomes["error"] == 1504


def test_device_path_of_generate_and_verify_proof_refuses_a_bad_row_before_any_gpu_work():
    """host/prove_on_device.hpp GenerateAndVerifyProofOnDevice: a row that does not decode ends in stage 1 (decode) — checked here without a GPU,
    with null handles: nothing of the device path may have been touched when the row is bad"""
    import ctypes
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = ctypes.CDLL(os.path.join(root, "tests", "hostlib", "libdispatch_gpu.so"))
    host = ctypes.CDLL(os.path.join(root, "zkmerkle-proof-of-solvency_amd", "libzkpor_host.so"))
    host.zkh_witness_synth_encode.restype = ctypes.c_long
    buf = ctypes.create_string_buffer(1 << 22)
    n = host.zkh_witness_synth_encode(ctypes.c_uint64(3), 2, 3, 500, 1, 2, buf, ctypes.c_size_t(1 << 22))
    assert n > 0
    column = buf.raw[:n]
    drv.prove_row_on_device.restype = ctypes.c_long
    r = (ctypes.c_uint64 * 4)(1, 0, 0, 0); s = (ctypes.c_uint64 * 4)(2, 0, 0, 0)
    for bad in (column[: (len(column) // 2) & ~3], b"!!!!" + column[4:], column[:-3]):
        err = ctypes.create_string_buffer(256)
        rc = drv.prove_row_on_device(None, None, None, None, bad, ctypes.c_size_t(len(bad)), ctypes.c_int64(1), r, s, 0, ctypes.create_string_buffer(64), ctypes.c_size_t(64),
                                     ctypes.byref(ctypes.c_int()), (ctypes.c_uint8 * 512)(), ctypes.byref(ctypes.c_size_t()), (ctypes.c_uint8 * 256)(), (ctypes.c_uint8 * 32)(),
                                     None, None, err, ctypes.c_size_t(256))
        assert rc == -1 and err.value.startswith(b"decode:")
