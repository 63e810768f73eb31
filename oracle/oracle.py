"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes loader for oracle/liboracle.so (CPU restatement of the
reference's hot path).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
Arrays are numpy uint64 with a trailing dimension of limbs (Fr/Fp: 4, G1 affine: 8, G2 affine: 16)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("capi.cpp", "algos.hpp", "bn254.hpp", "poseidon.hpp", "pairing.hpp", "pairing_consts.inc", "marshal.hpp")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_synth_create.restype = ctypes.c_void_p
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
P_MOD = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47


def ints_to_limbs(vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(a[i, j]) << (64 * j) for j in range(4)) for i in range(a.shape[0])]


def fr_from_ints(vals):
    c = ints_to_limbs([v % R_MOD for v in vals])
    out = np.empty_like(c)
    lib().orc_fr_from_canon(_p(c), _p(out), ctypes.c_size_t(len(vals)))
    return out


def fr_to_ints(a):
    a = _u64(a).reshape(-1, 4)
    out = np.empty_like(a)
    lib().orc_fr_to_canon(_p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    return limbs_to_ints(out)


def fp_from_ints(vals):
    c = ints_to_limbs([v % P_MOD for v in vals])
    out = np.empty_like(c)
    lib().orc_fp_from_canon(_p(c), _p(out), ctypes.c_size_t(len(vals)))
    return out


def fp_to_ints(a):
    a = _u64(a).reshape(-1, 4)
    out = np.empty_like(a)
    lib().orc_fp_to_canon(_p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    return limbs_to_ints(out)


def fr_random(seed, n):
    out = np.empty((n, 4), dtype=np.uint64)
    lib().orc_fr_random(ctypes.c_uint64(seed), _p(out), ctypes.c_size_t(n))
    return out


def _binop(name, a, b):
    a = _u64(a); b = _u64(b)
    out = np.empty_like(a)
    getattr(lib(), name)(_p(a), _p(b), _p(out), ctypes.c_size_t(a.reshape(-1, 4).shape[0]))
    return out


def fp_mul(a, b): return _binop("orc_fp_mul", a, b)
def fp_add(a, b): return _binop("orc_fp_add", a, b)
def fp_sub(a, b): return _binop("orc_fp_sub", a, b)
def fr_mul(a, b): return _binop("orc_fr_mul", a, b)
def fr_add(a, b): return _binop("orc_fr_add", a, b)
def fr_sub(a, b): return _binop("orc_fr_sub", a, b)


def fr_inv(a):
    a = _u64(a); out = np.empty_like(a)
    lib().orc_fr_inv(_p(a), _p(out), ctypes.c_size_t(a.reshape(-1, 4).shape[0]))
    return out


def fp_inv(a):
    a = _u64(a); out = np.empty_like(a)
    lib().orc_fp_inv(_p(a), _p(out), ctypes.c_size_t(a.reshape(-1, 4).shape[0]))
    return out


def fr_dot(a, b):
    a = _u64(a); b = _u64(b)
    out = np.empty((4,), dtype=np.uint64)
    lib().orc_fr_dot(_p(a), _p(b), ctypes.c_size_t(a.shape[0]), _p(out))
    return out


def g1_from_scalars(sc):
    sc = _u64(sc); n = sc.shape[0]
    out = np.empty((n, 8), dtype=np.uint64)
    lib().orc_g1_from_scalars(_p(sc), ctypes.c_size_t(n), _p(out))
    return out


def g2_from_scalars(sc):
    sc = _u64(sc); n = sc.shape[0]
    out = np.empty((n, 16), dtype=np.uint64)
    lib().orc_g2_from_scalars(_p(sc), ctypes.c_size_t(n), _p(out))
    return out


def g1_on_curve(p):
    p = _u64(p); return bool(lib().orc_g1_on_curve(_p(p), ctypes.c_size_t(p.reshape(-1, 8).shape[0])))


def g2_on_curve(p):
    p = _u64(p); return bool(lib().orc_g2_on_curve(_p(p), ctypes.c_size_t(p.reshape(-1, 16).shape[0])))


def g1_jac_to_affine(j):
    j = _u64(j).reshape(-1, 12); out = np.empty((j.shape[0], 8), dtype=np.uint64)
    lib().orc_g1_jac_to_affine(_p(j), _p(out), ctypes.c_size_t(j.shape[0]))
    return out


def g2_jac_to_affine(j):
    j = _u64(j).reshape(-1, 24); out = np.empty((j.shape[0], 16), dtype=np.uint64)
    lib().orc_g2_jac_to_affine(_p(j), _p(out), ctypes.c_size_t(j.shape[0]))
    return out


def g1_xyzz_to_affine(x):
    x = _u64(x).reshape(-1, 16); out = np.empty((x.shape[0], 8), dtype=np.uint64)
    lib().orc_g1_xyzz_to_affine(_p(x), _p(out), ctypes.c_size_t(x.shape[0]))
    return out


def g1_add(a, b):
    a = _u64(a).reshape(-1, 8); b = _u64(b).reshape(-1, 8); out = np.empty_like(a)
    lib().orc_g1_add_affine(_p(a), _p(b), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def g2_add(a, b):
    a = _u64(a).reshape(-1, 16); b = _u64(b).reshape(-1, 16); out = np.empty_like(a)
    lib().orc_g2_add_affine(_p(a), _p(b), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def g1_scalar_mul(p, k):
    p = _u64(p).reshape(-1, 8); k = _u64(k).reshape(-1, 4); out = np.empty_like(p)
    lib().orc_g1_scalar_mul(_p(p), _p(k), _p(out), ctypes.c_size_t(p.shape[0]))
    return out


def g1_msm(pts, sc, window=0):
    pts = _u64(pts); sc = _u64(sc); out = np.empty((8,), dtype=np.uint64)
    lib().orc_g1_msm(_p(pts), _p(sc), ctypes.c_size_t(sc.shape[0]), ctypes.c_int(window), _p(out))
    return out


def g2_msm(pts, sc, window=0):
    pts = _u64(pts); sc = _u64(sc); out = np.empty((16,), dtype=np.uint64)
    lib().orc_g2_msm(_p(pts), _p(sc), ctypes.c_size_t(sc.shape[0]), ctypes.c_int(window), _p(out))
    return out


DIT, DIF = 0, 1


def fft(a, log2n, inverse=False, decimation=DIF, on_coset=False):
    a = _u64(a).copy()
    lib().orc_fft(_p(a), ctypes.c_int(log2n), ctypes.c_int(int(inverse)), ctypes.c_int(decimation),
                  ctypes.c_int(int(on_coset)))
    return a


def bit_reverse(a, log2n):
    a = _u64(a).copy()
    lib().orc_bit_reverse(_p(a), ctypes.c_int(log2n))
    return a


def dft_naive(a, log2n, on_coset=False):
    a = _u64(a); out = np.empty_like(a)
    lib().orc_dft_naive(_p(a), ctypes.c_int(log2n), ctypes.c_int(int(on_coset)), _p(out))
    return out


def compute_h(a, b, c, log2d):
    a = _u64(a); b = _u64(b); c = _u64(c)
    out = np.empty((1 << log2d, 4), dtype=np.uint64)
    lib().orc_compute_h(_p(a), _p(b), _p(c), ctypes.c_size_t(a.shape[0]), ctypes.c_int(log2d), _p(out))
    return out


def quotient_identity(log2d, a, b, c, h, tau, h_bitrev=True, want_values=False):
    """FFT-free check of computeH's output (oracle/quotient.hpp): H(tau) (tau^D - 1) == A(tau) B(tau) - C(tau) with A, B, C evaluated
    from their values on the domain by barycentric Lagrange sums and H from its coefficient vector h (bit-reversed order = what
    computeH returns).  a, b, c: n_cons <= D rows (zero padded); h: D rows; tau: one Fr (Montgomery limbs).  True iff it holds."""
    a = _u64(a).reshape(-1, 4); b = _u64(b).reshape(-1, 4); c = _u64(c).reshape(-1, 4); h = _u64(h).reshape(-1, 4)
    assert a.shape == b.shape == c.shape and a.shape[0] <= (1 << log2d) and h.shape[0] == (1 << log2d)
    t = _u64(tau).reshape(4)
    out = np.empty((6, 4), dtype=np.uint64)
    rc = lib().orc_quotient_identity(ctypes.c_int(log2d), _p(a), _p(b), _p(c), ctypes.c_size_t(a.shape[0]), _p(h), ctypes.c_int(int(h_bitrev)),
                                     _p(t), _p(out))
    if rc < 0:
        raise ValueError("tau lies in the evaluation domain")
    return (bool(rc), out) if want_values else bool(rc)


def poseidon_set_convention(out_idx, carry_idx):
    lib().orc_poseidon_set_convention(ctypes.c_int(out_idx), ctypes.c_int(carry_idx))


def poseidon_params(t):
    rp = lib().orc_poseidon_rp(ctypes.c_int(t))
    rc = np.empty(((8 + rp) * t, 4), dtype=np.uint64)
    mds = np.empty((t * t, 4), dtype=np.uint64)
    lib().orc_poseidon_params(ctypes.c_int(t), _p(rc), _p(mds))
    return rp, rc, mds


def poseidon_permute(state):
    s = _u64(state).copy()
    lib().orc_poseidon_permute(_p(s), ctypes.c_int(s.shape[0]))
    return s


def poseidon_permute_trace(states, t):
    """(final states, trace) of len(states) permutations of width t; trace[(s * 3 + c), i] = x^2 / x^4 / x^5 of S-box s of permutation i
    (the slot order of zkpor_witgen_poseidon_trace_dev)"""
    st = _u64(states).reshape(-1, t, 4).copy()
    n = st.shape[0]
    ns = 8 * t + int(lib().orc_poseidon_rp(ctypes.c_int(t)))
    tr = np.empty((3 * ns, n, 4), dtype=np.uint64)
    lib().orc_poseidon_permute_trace(_p(st), ctypes.c_int(t), ctypes.c_size_t(n), _p(tr))
    return st, tr


def poseidon_hash(inp):
    inp = _u64(inp); out = np.empty((4,), dtype=np.uint64)
    lib().orc_poseidon_hash(_p(inp), ctypes.c_size_t(inp.shape[0]), _p(out))
    return out


def poseidon_hash2_batch(pairs):
    pairs = _u64(pairs).reshape(-1, 4)
    n = pairs.shape[0] // 2
    out = np.empty((n, 4), dtype=np.uint64)
    lib().orc_poseidon_hash2_batch(_p(pairs), ctypes.c_size_t(n), _p(out))
    return out


ACCOUNT_DTYPE = np.dtype([("id_be", np.uint8, 32), ("equity", np.uint64, 2), ("debt", np.uint64, 2),
                          ("collateral", np.uint64, 2), ("n_assets", np.uint32), ("asset_off", np.uint32)])
ASSET_DTYPE = np.dtype([("equity", np.uint64), ("debt", np.uint64), ("loan", np.uint64), ("margin", np.uint64),
                        ("portfolio_margin", np.uint64), ("index", np.uint32), ("pad", np.uint32)])
assert ACCOUNT_DTYPE.itemsize == 88 and ASSET_DTYPE.itemsize == 48


def account_leaves(accounts, assets, tier):
    accounts = np.ascontiguousarray(accounts, dtype=ACCOUNT_DTYPE)
    assets = np.ascontiguousarray(assets, dtype=ASSET_DTYPE)
    out = np.empty((accounts.shape[0], 4), dtype=np.uint64)
    lib().orc_account_leaves(_p(accounts), _p(assets), ctypes.c_size_t(accounts.shape[0]), ctypes.c_int(tier), _p(out))
    return out


TIER_DTYPE = np.dtype([("boundary", np.uint64, 2), ("ratio", np.uint8), ("pad", np.uint8, 7)])
CEX_CONST_DTYPE = np.dtype([("base_price", np.uint64), ("loan", TIER_DTYPE, 12), ("margin", TIER_DTYPE, 12), ("portfolio_margin", TIER_DTYPE, 12)])
CEX_TOTALS_DTYPE = np.dtype([("total_equity", np.uint64), ("total_debt", np.uint64), ("loan_collateral", np.uint64),
                             ("margin_collateral", np.uint64), ("portfolio_margin_collateral", np.uint64)])
assert TIER_DTYPE.itemsize == 24 and CEX_CONST_DTYPE.itemsize == 872 and CEX_TOTALS_DTYPE.itemsize == 40


def cex_commitments(consts, totals):
    """utils.ComputeCexAssetsCommitment for every row of totals[n_states, n_assets]"""
    consts = np.ascontiguousarray(consts, dtype=CEX_CONST_DTYPE); totals = np.ascontiguousarray(totals, dtype=CEX_TOTALS_DTYPE)
    n_assets = consts.shape[0]; n_states = totals.size // n_assets
    out = np.empty((n_states, 4), dtype=np.uint64)
    lib().orc_cex_commitments(_p(consts), ctypes.c_size_t(n_assets), _p(totals), ctypes.c_size_t(n_states), _p(out))
    return out


def tier_query(tiers, value):
    """(index, flag, collateral value, precomputed[]) for one tier list [(boundary, ratio), ...] and an integer value"""
    n = len(tiers)
    t3 = np.zeros((n, 3), dtype=np.uint64)
    for i, (b, r) in enumerate(tiers):
        t3[i] = (b & ((1 << 64) - 1), b >> 64, r)
    v = np.array([value & ((1 << 64) - 1), value >> 64], dtype=np.uint64)
    idx = ctypes.c_int(); flag = ctypes.c_int(); out = np.zeros(2, dtype=np.uint64); pre = np.zeros((n, 2), dtype=np.uint64)
    lib().orc_tier_query(_p(t3), ctypes.c_int(n), _p(v), ctypes.byref(idx), ctypes.byref(flag), _p(out), _p(pre))
    return idx.value, flag.value, int(out[0]) | (int(out[1]) << 64), [int(a) | (int(b) << 64) for a, b in pre]


def account_totals(accounts, assets, consts):
    """fills equity / debt / collateral of a copy of `accounts`; returns (accounts, valid[n])"""
    accounts = np.ascontiguousarray(accounts, dtype=ACCOUNT_DTYPE).copy(); assets = np.ascontiguousarray(assets, dtype=ASSET_DTYPE)
    consts = np.ascontiguousarray(consts, dtype=CEX_CONST_DTYPE)
    valid = np.zeros(accounts.shape[0], dtype=np.uint8)
    lib().orc_account_totals(_p(accounts), _p(assets), ctypes.c_size_t(accounts.shape[0]), _p(consts), ctypes.c_size_t(consts.shape[0]), _p(valid))
    return accounts, valid


def merkle_build(leaves, depth, nil_leaf, want_levels=False):
    leaves = _u64(leaves).reshape(-1, 4); n = leaves.shape[0]
    nil_leaf = _u64(nil_leaf)
    tot = sum((n + (1 << l) - 1) >> l for l in range(1, depth + 1))
    levels = np.empty((tot, 4), dtype=np.uint64) if want_levels else None
    nil = np.empty((depth + 1, 4), dtype=np.uint64)
    root = np.empty((4,), dtype=np.uint64)
    lib().orc_merkle_build(_p(leaves), ctypes.c_size_t(n), ctypes.c_int(depth), _p(nil_leaf),
                           _p(levels) if want_levels else None, _p(nil), _p(root))
    return root, nil, levels


def sparse_tree(keys, leaves, depth, nil_leaf, query_keys):
    """(root, proofs[nq, depth, 4]) of the tree with `leaves` Set at `keys` (merkletree.go Set/Build/GetProof)"""
    keys = np.ascontiguousarray(keys, dtype=np.uint32); leaves = _u64(leaves).reshape(-1, 4)
    q = np.ascontiguousarray(query_keys, dtype=np.uint32)
    proofs = np.empty((q.shape[0], depth, 4), dtype=np.uint64)
    root = np.empty((4,), dtype=np.uint64)
    lib().orc_sparse_tree(_p(keys), _p(leaves), ctypes.c_size_t(keys.shape[0]), ctypes.c_int(depth), _p(_u64(nil_leaf)),
                          _p(q), ctypes.c_size_t(q.shape[0]), _p(proofs), _p(root))
    return root, proofs


def merkle_verify(root, key, proof, leaf):
    proof = _u64(proof).reshape(-1, 4)
    return bool(lib().orc_merkle_verify(_p(_u64(root)), ctypes.c_uint32(key), _p(proof), ctypes.c_int(proof.shape[0]), _p(_u64(leaf))))


def fr_to_be(a):
    a = _u64(a).reshape(-1, 4); out = np.empty((a.shape[0], 32), dtype=np.uint8)
    lib().orc_fr_to_be(_p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def fr_from_be(b):
    b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 32); out = np.empty((b.shape[0], 4), dtype=np.uint64)
    lib().orc_fr_from_be(_p(b), _p(out), ctypes.c_size_t(b.shape[0]))
    return out


class Synth:
    """trapdoor-known synthetic Groth16 instance + key (oracle/algos.hpp synth_*)"""

    def __init__(self, n_inputs, n_cons, n_public=2, seed=1, z_bitrev=True):
        L = lib()
        self.h = ctypes.c_void_p(L.orc_synth_create(ctypes.c_size_t(n_inputs), ctypes.c_size_t(n_cons),
                                                    ctypes.c_size_t(n_public), ctypes.c_uint64(seed),
                                                    ctypes.c_int(int(z_bitrev))))
        dims = np.zeros(5, dtype=np.uint64)
        L.orc_synth_dims(self.h, _p(dims))
        self.log2d, self.n_wires, self.n_public, self.n_cons, self.n_z = (int(x) for x in dims)
        self.z_bitrev = z_bitrev
        nw = self.n_wires
        self.A = np.empty((nw, 8), np.uint64); self.B1 = np.empty((nw, 8), np.uint64)
        self.B2 = np.empty((nw, 16), np.uint64); self.K = np.empty((nw, 8), np.uint64)
        self.Z = np.empty((self.n_z, 8), np.uint64)
        self.abd1 = np.empty((3, 8), np.uint64); self.bd2 = np.empty((2, 16), np.uint64)
        self.w = np.empty((nw, 4), np.uint64)
        self.a = np.empty((self.n_cons, 4), np.uint64); self.b = np.empty_like(self.a); self.c = np.empty_like(self.a)
        L.orc_synth_export(self.h, _p(self.A), _p(self.B1), _p(self.B2), _p(self.K), _p(self.Z), _p(self.abd1),
                           _p(self.bd2), _p(self.w), _p(self.a), _p(self.b), _p(self.c))

    def prove_tail(self, r, s):
        out = np.empty(256, dtype=np.uint8)
        lib().orc_synth_prove_tail(self.h, _p(_u64(r)), _p(_u64(s)), _p(out))
        return out

    def check(self, r, s, proof256):
        proof256 = np.ascontiguousarray(proof256, dtype=np.uint8)
        return bool(lib().orc_synth_check(self.h, _p(_u64(r)), _p(_u64(s)), _p(proof256)))

    def r1cs(self):
        """the instance's constraint system in the product's layout: (coeff_table[n_coeff,4], [(row_ptr, coeff_ids, wire_ids)] x 3)"""
        L = lib()
        L.orc_synth_r1cs.restype = ctypes.c_size_t
        mats = []; all_coeffs = []
        for which in range(3):
            nnz = L.orc_synth_r1cs(self.h, ctypes.c_int(which), None, None, None)
            row_ptr = np.empty(self.n_cons + 1, dtype=np.uint64); wid = np.empty(nnz, dtype=np.uint32); co = np.empty((nnz, 4), dtype=np.uint64)
            L.orc_synth_r1cs(self.h, ctypes.c_int(which), _p(row_ptr), _p(wid), _p(co))
            mats.append((row_ptr, wid, co)); all_coeffs.append(co)
        table, inv = np.unique(np.concatenate(all_coeffs), axis=0, return_inverse=True)
        inv = inv.reshape(-1).astype(np.uint32)
        out = []; off = 0
        for row_ptr, wid, co in mats:
            out.append((row_ptr, inv[off:off + co.shape[0]].copy(), wid)); off += co.shape[0]
        return np.ascontiguousarray(table), out

    def commitment_basis(self, committed_idx, sigma):
        """(Basis, BasisExpSigma) of a BSB22 commitment over the wires `committed_idx` for this setup (gamma-divided K values)"""
        ci = np.ascontiguousarray(committed_idx, dtype=np.uint32); sigma = _u64(sigma)
        basis = np.empty((ci.size, 8), dtype=np.uint64); bs = np.empty((ci.size, 8), dtype=np.uint64)
        lib().orc_synth_commitment_basis(self.h, _p(ci), ctypes.c_size_t(ci.size), _p(sigma), _p(basis), _p(bs))
        return basis, bs

    def verify_pairing_commit(self, proof256, commitment, pok, g2_sigma):
        """groth16.Verify's equation with one commitment: D joins the public-input sum, and the knowledge proof must hold"""
        proof256 = np.ascontiguousarray(proof256, dtype=np.uint8)
        return bool(lib().orc_synth_verify_pairing_commit(self.h, _p(proof256), _p(_u64(commitment)), _p(_u64(pok)), _p(_u64(g2_sigma))))

    def verify_pairing(self, proof256):
        """groth16.Verify's equation with a real pairing; uses only the vk, the public wires and the proof"""
        proof256 = np.ascontiguousarray(proof256, dtype=np.uint8)
        return bool(lib().orc_synth_verify_pairing(self.h, _p(proof256)))

    def __del__(self):
        try:
            lib().orc_synth_destroy(self.h)
        except Exception:
            pass


def r1cs_failing_rows(coeff_table, mats, w):
    """(failing rows, the lowest one or None) of (L w) o (R w) = O w — mats = [(row_ptr u64, coeff_ids u32, wire_ids u32)] for L, R, O, Montgomery limbs
    (oracle/capi.cpp orc_r1cs_failing_rows: the statement only, the oracle's own field arithmetic)"""
    co = _u64(coeff_table).reshape(-1, 4); wv = _u64(w).reshape(-1, 4)
    rp = [np.ascontiguousarray(m[0], dtype=np.uint64) for m in mats]
    ci = [np.ascontiguousarray(m[1], dtype=np.uint32) for m in mats]
    wi = [np.ascontiguousarray(m[2], dtype=np.uint32) for m in mats]
    n = rp[0].shape[0] - 1
    assert all(x.shape[0] == n + 1 for x in rp)
    P3 = ctypes.c_void_p * 3
    first = ctypes.c_uint64()
    L = lib()
    L.orc_r1cs_failing_rows.restype = ctypes.c_uint64
    bad = L.orc_r1cs_failing_rows(_p(co), ctypes.c_uint64(co.shape[0]), P3(*[x.ctypes.data for x in rp]), P3(*[x.ctypes.data for x in ci]), P3(*[x.ctypes.data for x in wi]),
                                  ctypes.c_uint64(n), _p(wv), ctypes.c_uint64(wv.shape[0]), ctypes.byref(first))
    return int(bad), (int(first.value) if bad else None)


def pairing(P, Q):
    """reduced Tate pairing t(P, Q) as 12 Fp (oracle/pairing.hpp)"""
    out = np.empty((12, 4), dtype=np.uint64)
    lib().orc_pairing(_p(_u64(P)), _p(_u64(Q)), _p(out))
    return out


def fp12_mul(a, b):
    out = np.empty((12, 4), dtype=np.uint64)
    lib().orc_fp12_mul(_p(_u64(a)), _p(_u64(b)), _p(out))
    return out


def fp12_pow_fr(a, e):
    out = np.empty((12, 4), dtype=np.uint64)
    lib().orc_fp12_pow_fr(_p(_u64(a)), _p(_u64(e)), _p(out))
    return out


def g2_mul_gen(k):
    out = np.empty(16, dtype=np.uint64)
    lib().orc_g2_mul_gen(_p(_u64(k)), _p(out))
    return out


def pedersen_verify_pairing(commitment, pok, g2_sigma):
    return bool(lib().orc_pedersen_verify_pairing(_p(_u64(commitment)), _p(_u64(pok)), _p(_u64(g2_sigma))))


def g1_compress(pts):
    pts = _u64(pts).reshape(-1, 8); out = np.empty((pts.shape[0], 32), dtype=np.uint8)
    lib().orc_g1_compress(_p(pts), ctypes.c_size_t(pts.shape[0]), _p(out))
    return out


def g2_compress(pts):
    pts = _u64(pts).reshape(-1, 16); out = np.empty((pts.shape[0], 64), dtype=np.uint8)
    lib().orc_g2_compress(_p(pts), ctypes.c_size_t(pts.shape[0]), _p(out))
    return out


def g1_decompress(b):
    """(rc, points): rc 0 ok, 1 not compressed, 2 out of range, 3 not on the curve (first offender)"""
    b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 32); out = np.zeros((b.shape[0], 8), dtype=np.uint64)
    rc = lib().orc_g1_decompress(_p(b), ctypes.c_size_t(b.shape[0]), _p(out))
    return rc, out


def g2_decompress(b):
    b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1, 64); out = np.zeros((b.shape[0], 16), dtype=np.uint64)
    rc = lib().orc_g2_decompress(_p(b), ctypes.c_size_t(b.shape[0]), _p(out))
    return rc, out


def proof_raw(proof256):
    proof256 = np.ascontiguousarray(proof256, dtype=np.uint8)
    out = np.empty(256, dtype=np.uint8)
    lib().orc_proof_raw(_p(proof256), _p(out))
    return out


def selftest():
    return lib().orc_selftest()


# ---- the CPU baseline port (cpubase.hpp): used by bench.py's cpu_baseline leg, checked against the plain oracle in tests ----
def fast_g1_msm(pts, sc, window=0):
    pts = _u64(pts); sc = _u64(sc)
    out = np.empty((8,), dtype=np.uint64)
    lib().orc_fast_g1_msm(_p(pts), _p(sc), ctypes.c_size_t(sc.shape[0]), ctypes.c_int(window), _p(out))
    return out


def fast_g2_msm(pts, sc, window=0):
    pts = _u64(pts); sc = _u64(sc)
    out = np.empty((16,), dtype=np.uint64)
    lib().orc_fast_g2_msm(_p(pts), _p(sc), ctypes.c_size_t(sc.shape[0]), ctypes.c_int(window), _p(out))
    return out


def fast_compute_h(a, b, c, log2d):
    """a, b, c: n_cons rows; returns h (2^log2d rows, the order compute_h returns)"""
    n = 1 << log2d
    bufs = []
    for v in (a, b, c):
        v = _u64(v)
        p = np.zeros((n, 4), dtype=np.uint64)
        p[: v.shape[0]] = v
        bufs.append(p)
    lib().orc_fast_compute_h(_p(bufs[0]), _p(bufs[1]), _p(bufs[2]), ctypes.c_int(log2d))
    return bufs[0]


def fast_prove_tail_work(log2d, g1, g2, w, a, b, c, n_commit):
    """one prove tail's worth of CPU work on all cores; a, b, c are overwritten.  Returns (fft_s, g1_s, g2_s, commit_s)"""
    times = np.zeros(4, dtype=np.float64)
    o1 = np.empty(8, dtype=np.uint64); o2 = np.empty(16, dtype=np.uint64)
    lib().orc_fast_prove_tail_work(ctypes.c_int(log2d), _p(g1), _p(g2), _p(w), _p(a), _p(b), _p(c), ctypes.c_size_t(n_commit), _p(times), _p(o1), _p(o2))
    return tuple(float(t) for t in times)


def threads():
    return int(lib().orc_threads())


def set_threads(n):
    """OpenMP thread count of the baseline port (libgomp of this process)"""
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))


def usable_cpus():
    """CPUs this process can really use: min(logical CPUs, affinity mask, cgroup CPU quota) and where the number came from.
    A container with cpu.max = "1600000 100000" gets 16 CPUs' worth of time however many it sees (more threads only get throttled)."""
    import math
    import os
    n = os.cpu_count() or 1
    why = f"{n} logical CPUs"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, why = a, f"affinity mask of {a} CPUs"
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]            # cgroup v2
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:                                                                      # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n, why = max(1, int(math.floor(quota + 1e-9))), f"cgroup CPU quota of {quota:g} CPUs (of {os.cpu_count()} logical)"
    return n, why
