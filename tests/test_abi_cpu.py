"""CPU suite: the C-ABI shared library loads without a GPU and exports every symbol include/zkpor.h declares; with no
usable device the product fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd", "libzkpor.so")


def _declared():
    hdr = open(os.path.join(ROOT, "include", "zkpor.h")).read()
    return sorted(set(re.findall(r"\b(zkpor_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(LIB)
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_abi_version_is_exported_and_checked_by_the_binding(monkeypatch):
    """include/zkpor.h ZKPOR_ABI_VERSION = the library's zkpor_abi_version() = the value zkpor.py was written against; a binding written
    against another value refuses to load the library (signatures changed in the middle of an argument list)"""
    import zkpor
    hdr = open(os.path.join(ROOT, "include", "zkpor.h")).read()
    declared = int(re.search(r"#define\s+ZKPOR_ABI_VERSION\s+(\d+)u", hdr).group(1))
    lib = ctypes.CDLL(LIB)
    lib.zkpor_abi_version.restype = ctypes.c_uint32
    assert lib.zkpor_abi_version() == declared == zkpor.ABI_VERSION
    go = open(os.path.join(ROOT, "go", "zkporgpu", "abi.go")).read()
    assert int(re.search(r"const abiVersion = (\d+)", go).group(1)) == declared
    monkeypatch.setattr(zkpor, "_lib", None)
    monkeypatch.setattr(zkpor, "ABI_VERSION", declared + 1)
    with pytest.raises(zkpor.ZkporError, match="ABI version"):
        zkpor.load_library()
    monkeypatch.setattr(zkpor, "ABI_VERSION", declared)
    monkeypatch.setattr(zkpor, "_lib", None)
    zkpor.load_library()


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import zkpor
    with pytest.raises(zkpor.ZkporError):
        zkpor.Context(0)


def test_proof_write_raw_is_host_only_and_sized():
    import numpy as np
    import zkpor
    proof = np.zeros(256, dtype=np.uint8)
    raw = zkpor.proof_write_raw(proof)
    assert raw.size == 324
    raw = zkpor.proof_write_raw(proof, np.zeros((1, 8), np.uint64), np.zeros(8, np.uint64))
    assert raw.size == 388 and raw[259] == 1   # the reference's 388-byte proof (one commitment), SURVEY.md a6.7
    # every point of this proof is the identity: gnark-crypto's RawBytes writes the flag mUncompressedInfinity (0b01 << 6) into
    # byte 0 and zeros behind it — not 64 zero bytes (marshal.go); Ar | Bs | Krs | count | commitment | knowledge proof
    expect = np.zeros(388, dtype=np.uint8)
    expect[[0, 64, 192, 260, 324]] = 0x40
    expect[259] = 1
    assert np.array_equal(raw, expect)
    import oracle as O
    assert np.array_equal(O.proof_raw(proof), expect[:256])
    # a real point keeps its two flag bits clear: (1, 2) is on the curve
    p1 = np.zeros(32, dtype=np.uint64); p1[0:4] = O.fp_from_ints([1])[0]; p1[4:8] = O.fp_from_ints([2])[0]
    raw = zkpor.proof_write_raw(p1.view(np.uint8))
    assert raw[0] == 0 and raw[31] == 1 and raw[63] == 2 and raw[64] == 0x40


def test_prove_assemble_matches_the_groth16_formulas_on_the_oracle():
    """zkpor_prove_assemble (host only): Ar = alpha + A.w + r delta, Bs = beta2 + B2.w + s delta2,
    Krs = K.w + Z.h + s Ar + r Bs1 - rs delta, from Jacobian sums in any projective form (one of them the point at infinity),
    against the same linear combinations evaluated by the oracle's multi-exponentiation"""
    import numpy as np
    import oracle as O
    import zkpor
    one = O.fp_from_ints([1])[0]
    g1 = O.g1_from_scalars(O.fr_random(51, 7))          # alpha, beta, delta, A.w, B1.w, K.w, Z.h
    g2 = O.g2_from_scalars(O.fr_random(52, 3))          # beta2, delta2, B2.w
    ones = O.fr_from_ints([1, 1])
    for trial, zero_slot in enumerate((None, "A", "Z", "B2")):
        r = O.fr_random(60 + trial, 1)[0]; s = O.fr_random(70 + trial, 1)[0]
        A, B1, K, Z = (g1[i].copy() for i in (3, 4, 5, 6)); B2 = g2[2].copy()
        if zero_slot == "A": A[:] = 0
        if zero_slot == "Z": Z[:] = 0
        if zero_slot == "B2": B2[:] = 0
        # Jacobian inputs in a NON-trivial projective form: (X l^2, Y l^3, l)
        def jac1(aff, seed):
            if not aff.any():
                return np.zeros(12, np.uint64)
            l = O.fp_from_ints([seed])[0]; l2 = O.fp_mul(l[None], l[None])[0]; l3 = O.fp_mul(l2[None], l[None])[0]
            return np.concatenate([O.fp_mul(aff[None, :4], l2[None])[0], O.fp_mul(aff[None, 4:], l3[None])[0], l])
        def jac2(aff):
            return np.concatenate([aff, one, np.zeros(4, np.uint64)]) if aff.any() else np.zeros(24, np.uint64)
        sums = np.concatenate([jac1(A, 3), jac1(B1, 5), jac2(B2), jac1(K, 7), jac1(Z, 11)]).view(np.uint8)
        proof = zkpor.prove_assemble((g1[0], g1[1], g1[2], g2[0], g2[1]), sums, r, s).view(np.uint64)
        rs_neg = O.fr_sub(O.fr_from_ints([0]), O.fr_mul(r[None], s[None]))[0]
        lin1 = lambda pts, sc: O.g1_msm(np.stack(pts), np.stack(sc))
        one_fr = O.fr_from_ints([1])[0]
        ar = lin1([g1[0], A, g1[2]], [one_fr, one_fr, r])
        bs1 = lin1([g1[1], B1, g1[2]], [one_fr, one_fr, s])
        krs = lin1([K, Z, ar, bs1, g1[2]], [one_fr, one_fr, s, r, rs_neg])
        bs2 = O.g2_msm(np.stack([g2[0], B2, g2[1]]), np.stack([one_fr, one_fr, s]))
        assert np.array_equal(proof[0:8], ar) and np.array_equal(proof[8:24], bs2) and np.array_equal(proof[24:32], krs)


def test_jac_sum_edge_cases_host_only():
    """zkpor_g1/g2_jac_sum (the host-side reduction of the partial sums of a split proof): equal parts (doubling), opposite parts
    (infinity), infinity parts, a single part, no part"""
    import numpy as np
    import oracle as O
    import zkpor
    one = O.fp_from_ints([1])[0]
    P = O.g1_from_scalars(O.fr_random(81, 3))
    j1 = lambda a: np.concatenate([a, one]) if a.any() else np.zeros(12, np.uint64)
    neg = P[0].copy(); neg[4:] = O.fp_sub(O.fp_from_ints([0]), P[0][None, 4:])[0]
    aff = lambda parts: O.g1_jac_to_affine(zkpor.g1_jac_sum(np.stack(parts)))[0]
    two = O.fr_from_ints([2]); ones3 = O.fr_from_ints([1, 1, 1])
    assert np.array_equal(aff([j1(P[0]), j1(P[0])]), O.g1_msm(P[:1], two))                       # P + P
    assert not aff([j1(P[0]), j1(neg)]).any()                                                     # P - P
    assert np.array_equal(aff([j1(P[0]), j1(neg), j1(P[1])]), P[1])                               # through infinity and on
    assert np.array_equal(aff([np.zeros(12, np.uint64), j1(P[2]), np.zeros(12, np.uint64)]), P[2])
    assert np.array_equal(aff([j1(P[0]), j1(P[1]), j1(P[2])]), O.g1_msm(P, ones3))
    assert np.array_equal(aff([j1(P[1])]), P[1])
    assert not O.g1_jac_to_affine(zkpor.g1_jac_sum(np.zeros((0, 12), np.uint64)))[0].any()
    Q = O.g2_from_scalars(O.fr_random(82, 2))
    j2 = lambda a: np.concatenate([a, one, np.zeros(4, np.uint64)]) if a.any() else np.zeros(24, np.uint64)
    negq = Q[0].copy(); negq[8:] = np.concatenate([O.fp_sub(O.fp_from_ints([0]), Q[0][None, 8:12])[0], O.fp_sub(O.fp_from_ints([0]), Q[0][None, 12:16])[0]])
    aff2 = lambda parts: O.g2_jac_to_affine(zkpor.g2_jac_sum(np.stack(parts)))[0]
    assert np.array_equal(aff2([j2(Q[0]), j2(Q[0])]), O.g2_msm(Q[:1], two))
    assert not aff2([j2(Q[0]), j2(negq)]).any()
    assert np.array_equal(aff2([j2(Q[0]), j2(Q[1])]), O.g2_msm(Q, O.fr_from_ints([1, 1])))


def test_every_entry_point_stops_exceptions_at_the_abi():
    """include/zkpor.h: "they never throw" — every `int32_t zkpor_*` definition in csrc/*.hip is a function-try-block ending in
    ZK_ABI_CATCH, the void / double ones catch everything (VERDICT r04 weak #1b: a cgo caller cannot unwind; the reference's prover
    expects an error return, src/prover/prover/prover.go:269-272)"""
    import glob
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "abi_firewall.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    wrapped = 0
    for p in glob.glob(os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd", "csrc", "*.hip")):
        src = open(p).read()
        wrapped += src.count("ZK_ABI_CATCH")
        for m in re.finditer(r"^(void|double) (zkpor_\w+)\(.*$", src, re.M):
            assert m.group(0).rstrip().endswith("try {"), m.group(0)
    assert wrapped >= 100
