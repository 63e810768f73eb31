"""host/bsb22_challenge.hpp: the Fr challenge the BSB22 commitment hint returns to the solver =
gnark-crypto's fr.Hash (RFC 9380 expand_message_xmd over SHA-256, 48 bytes, mod r) of commitment.Marshal() || public
committed values under the DST "bsb22-commitment".

Pins: SHA-256 against hashlib; expand_message_xmd against the RFC 9380 appendix K.1 known answers and against an independent
restatement of the RFC's pseudo-code written here on hashlib; the reduction against Python integers."""
import ctypes
import hashlib
import os
import random

import pytest

from test_dispatcher_cpu import host  # noqa: F401  (the libzkpor_host.so fixture)

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def xmd_py(msg: bytes, dst: bytes, n: int) -> bytes:
    """RFC 9380 section 5.3.1, H = SHA-256 (b_in_bytes 32, s_in_bytes 64)."""
    ell = (n + 31) // 32
    assert ell <= 255 and n <= 65535 and len(dst) <= 255
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(b"\x00" * 64 + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime).digest()
    b = [hashlib.sha256(b0 + b"\x01" + dst_prime).digest()]
    for i in range(2, ell + 1):
        x = bytes(p ^ q for p, q in zip(b0, b[-1]))
        b.append(hashlib.sha256(x + bytes([i]) + dst_prime).digest())
    return b"".join(b)[:n]


def c_xmd(host, msg, dst, n):
    out = (ctypes.c_uint8 * n)()
    rc = host.zkh_expand_msg_xmd(msg, len(msg), dst, len(dst), out, n)
    return rc, bytes(out)


@pytest.fixture(autouse=True)
def _sigs(host):
    host.zkh_sha256.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    host.zkh_sha256.restype = None
    host.zkh_expand_msg_xmd.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    host.zkh_fr_hash.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
    host.zkh_bsb22_challenge.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]


def test_sha256_against_hashlib(host):
    rng = random.Random(5)
    for n in list(range(0, 130)) + [255, 256, 1000, 4096, 65537]:
        m = bytes(rng.randrange(256) for _ in range(n))
        out = (ctypes.c_uint8 * 32)()
        host.zkh_sha256(m, n, out)
        assert bytes(out) == hashlib.sha256(m).digest(), n


# RFC 9380 appendix K.1: expand_message_xmd(SHA-256), DST = "QUUX-V01-CS02-with-expander-SHA256-128"
RFC_DST = b"QUUX-V01-CS02-with-expander-SHA256-128"
RFC_K1 = [
    (b"", 0x20, "68a985b87eb6b46952128911f2a4412bbc302a9d759667f87f7a21d803f07235"),
    (b"abc", 0x20, "d8ccab23b5985ccea865c6c97b6e5b8350e794e603b4b97902f53a8a0d605615"),
    (b"abcdef0123456789", 0x20, "eff31487c770a893cfb36f912fbfcbff40d5661771ca4b2cb4eafe524333f5c1"),
]


def test_expand_message_xmd_rfc9380_known_answers(host):
    for msg, n, want in RFC_K1:
        assert xmd_py(msg, RFC_DST, n).hex() == want          # the restatement reproduces the RFC
        rc, got = c_xmd(host, msg, RFC_DST, n)
        assert rc == 0 and got.hex() == want                  # and so does the C++ one


def test_expand_message_xmd_against_restatement(host):
    rng = random.Random(9)
    for _ in range(200):
        msg = bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 31, 32, 33, 64, 65, 100, 500])))
        dst = bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 16, 38, 255])))
        n = rng.choice([1, 31, 32, 33, 48, 96, 255 * 32])
        rc, got = c_xmd(host, msg, dst, n)
        assert rc == 0 and got == xmd_py(msg, dst, n)
    # the RFC's limits are refusals, not truncations
    assert c_xmd(host, b"x", b"d" * 256, 32)[0] == 1
    assert c_xmd(host, b"x", b"d", 255 * 32 + 1)[0] == 1


def fr_hash_py(msg, dst, count):
    u = xmd_py(msg, dst, 48 * count)
    return [int.from_bytes(u[48 * i:48 * i + 48], "big") % R for i in range(count)]


def test_fr_hash_and_reduction(host):
    rng = random.Random(11)
    for _ in range(100):
        msg = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 200)))
        count = rng.randrange(1, 5)
        out = (ctypes.c_uint8 * (32 * count))()
        assert host.zkh_fr_hash(msg, len(msg), b"bsb22-commitment", 16, count, out) == 0
        got = [int.from_bytes(bytes(out)[32 * i:32 * i + 32], "big") for i in range(count)]
        assert got == fr_hash_py(msg, b"bsb22-commitment", count)
        assert all(g < R for g in got)


def test_bsb22_challenge_layout(host):
    """hash input = 64-byte commitment || 32-byte big-endian public committed values, DST "bsb22-commitment", one element."""
    rng = random.Random(13)
    for n_pub in (0, 1, 3):
        com = bytes(rng.randrange(256) for _ in range(64))
        pub = [rng.randrange(R) for _ in range(n_pub)]
        pub_b = b"".join(p.to_bytes(32, "big") for p in pub)
        out = (ctypes.c_uint8 * 32)()
        assert host.zkh_bsb22_challenge(com, pub_b, n_pub, out) == 0
        assert int.from_bytes(bytes(out), "big") == fr_hash_py(com + pub_b, b"bsb22-commitment", 1)[0]
