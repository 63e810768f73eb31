"""ORACLE — TEST INFRASTRUCTURE ONLY (tests/, smoke() and bench.py's untimed `checked` / cpu_baseline legs).

The trapdoor of the synthetic proving key (`zkpor_pk_synth`, csrc/groth16.hip `k_synth_points`): every point of every
array is s_i * G with a known integer s_i, so each of the sums groth16.Prove forms (gnark
backend/groth16/bn254/prove.go) is checkable in the exponent at ANY size with one dot product over Fr and one fixed-base
multiplication:

    Ar  = alpha + sum_i w_i A_i + r delta                     =  (k_alpha + <sA, w> + r k_delta) G1
    Bs  = beta2 + sum_i w_i B2_i + s delta2                    =  (k_beta  + <sB, w> + s k_delta) G2
    Krs = sum_i w_i K_i + sum_j h_j Z_j + s Ar + r Bs1 - rs delta
        = (<sK, w> + <sZ, h> + s ar + r bs - r s k_delta) G1
    commitment = <sCB, v> G1,  knowledge proof = <sCBS, v> G1  (gnark-crypto pedersen.ProvingKey.Commit / ProveKnowledge)

This file restates the generator of the scalars (SplitMix-style mixer, runs of 32 points k + j q) independently of the
device code, in numpy on 64-bit words; tests/test_fullsize_gpu.py checks it against the scalar Python form in zkpor.py.
"""
import numpy as np

import oracle as O

SYNTH_RUN = 32
SYNTH_Q = 0x9e3779b97f4a7c15
_M64 = (1 << 64) - 1
# infinity pattern of the synthetic arrays (csrc/groth16.hip zkpor_pk_synth): array id -> modulus of the hash test
G1_A, G1_B, G1_K, G1_Z, G1_COMMIT_BASIS, G1_COMMIT_BASIS_SIGMA = range(6)
INF_MOD = {G1_A: 64, G1_B: 10, G1_K: 4, G1_Z: 0, G1_COMMIT_BASIS: 0, G1_COMMIT_BASIS_SIGMA: 0}


def _smix(x):
    x = x + np.uint64(0x9e3779b97f4a7c15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
    return x ^ (x >> np.uint64(31))


def synth_k(seed, arr, run):
    """python-int form of the 64-bit run scalar (also the scalars of alpha / beta / delta: arr = 100 / 101 / 102, run 0)"""
    x = (seed ^ (((arr + 1) * 0xa0761d6478bd642f) & _M64) ^ ((run * 0xe7037ed1a0b428db) & _M64)) & _M64
    x = (x + 0x9e3779b97f4a7c15) & _M64
    x = ((x ^ (x >> 30)) * 0xbf58476d1ce4e5b9) & _M64
    x = ((x ^ (x >> 27)) * 0x94d049bb133111eb) & _M64
    return (x ^ (x >> 31)) | 1


def synth_scalars_canon(seed, arr, lo, hi, inf_mod=None, inf_below=0):
    """canonical limbs (hi-lo, 4) of s_i = k(i // 32) + (i % 32) q for i in [lo, hi), zero where the synthetic point is infinity"""
    if inf_mod is None:
        inf_mod = INF_MOD[arr]
    with np.errstate(over="ignore"):
        i = np.arange(lo, hi, dtype=np.uint64)
        run = i // np.uint64(SYNTH_RUN)
        j = (i % np.uint64(SYNTH_RUN)).astype(np.int64)
        x = np.uint64(seed & _M64) ^ np.uint64(((arr + 1) * 0xa0761d6478bd642f) & _M64) ^ (run * np.uint64(0xe7037ed1a0b428db))
        k = _smix(x) | np.uint64(1)
        jq = [jj * SYNTH_Q for jj in range(SYNTH_RUN)]
        jq_lo = np.array([v & _M64 for v in jq], dtype=np.uint64)[j]
        jq_hi = np.array([v >> 64 for v in jq], dtype=np.uint64)[j]
        s_lo = k + jq_lo
        s_hi = jq_hi + (s_lo < k).astype(np.uint64)
        out = np.zeros((hi - lo, 4), dtype=np.uint64)
        out[:, 0] = s_lo
        out[:, 1] = s_hi
        if inf_mod:
            h = i * np.uint64(0xd6e8feb86659fd93)
            h ^= h >> np.uint64(32)
            out[(h % np.uint64(inf_mod)) == 0] = 0
        if inf_below > lo:
            out[: inf_below - lo] = 0
    return out


def synth_dot(seed, arr, scalars_mont, inf_mod=None, inf_below=0, chunk=1 << 22, fused=True):
    """<s, x> over Fr (Montgomery limbs, shape (4,)) for the first len(x) points of synthetic array `arr`.  fused: one OpenMP pass in the
    oracle library (orc_synth_dot: generator and dot product restated in C); otherwise the numpy generator in chunks — the two are
    compared in tests/test_trapdoor_cpu.py, the fused one is what the 2^26 checks use (4 x 2^26 elements in ~1 s instead of ~12 s)"""
    x = O._u64(scalars_mont).reshape(-1, 4)
    n = x.shape[0]
    if fused:
        import ctypes
        out = np.empty(4, dtype=np.uint64)
        O.lib().orc_synth_dot(ctypes.c_uint64(seed & _M64), ctypes.c_int(arr), O._p(x), ctypes.c_size_t(n),
                              ctypes.c_uint64(INF_MOD[arr] if inf_mod is None else inf_mod), ctypes.c_size_t(inf_below), O._p(out))
        return out
    acc = O.fr_from_ints([0])[0]
    buf = np.empty((min(chunk, n), 4), dtype=np.uint64)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        c = synth_scalars_canon(seed, arr, lo, hi, inf_mod, inf_below)
        m = buf[: hi - lo]
        O.lib().orc_fr_from_canon(O._p(c), O._p(m), hi - lo)
        acc = O.fr_add(acc.reshape(1, 4), O.fr_dot(m, x[lo:hi]).reshape(1, 4))[0]
    return acc


def _fr(v):
    return O.fr_from_ints([v])[0]


def _mul(a, b):
    return O.fr_mul(a.reshape(1, 4), b.reshape(1, 4))[0]


def _add(a, b):
    return O.fr_add(a.reshape(1, 4), b.reshape(1, 4))[0]


def _sub(a, b):
    return O.fr_sub(a.reshape(1, 4), b.reshape(1, 4))[0]


class SynthKeyTrapdoor:
    """the discrete logs of one synthetic key against one (w, h): the four dot products are taken once, any number of
    proofs with different blinding (r, s) are then checked with a handful of Fr operations and three fixed-base products"""

    def __init__(self, seed, n_public, w_mont, h_mont, dZ=None, masks=None):
        """h_mont: the first len(Z) = D - 1 scalars of h in the key's order; dZ: <sZ, h> from another instance over the same h.
        masks = (inf_a, inf_b, removed_from_k) for a key made by zkpor_pk_synth_masked: byte masks / wire ids instead of the seeded infinity pattern"""
        w = O._u64(w_mont).reshape(-1, 4)
        self.k_alpha = _fr(synth_k(seed, 100, 0)); self.k_beta = _fr(synth_k(seed, 101, 0)); self.k_delta = _fr(synth_k(seed, 102, 0))
        if masks is not None:
            inf_a, inf_b, removed = masks

            def masked(m):
                x = w.copy(); x[np.asarray(m, dtype=bool)] = 0
                return x
            km = np.zeros(w.shape[0], dtype=bool); km[np.asarray(removed, dtype=np.int64)] = True
            self.dA = synth_dot(seed, G1_A, masked(inf_a), inf_mod=0)
            self.dB = synth_dot(seed, G1_B, masked(inf_b), inf_mod=0)
            self.dK = synth_dot(seed, G1_K, masked(km), inf_mod=0, inf_below=n_public)
        else:
            self.dA = synth_dot(seed, G1_A, w)
            self.dB = synth_dot(seed, G1_B, w)                       # B1 and B2 carry the same scalars
            self.dK = synth_dot(seed, G1_K, w, inf_below=n_public)
        self.dZ = dZ if dZ is not None else synth_dot(seed, G1_Z, O._u64(h_mont).reshape(-1, 4))   # h in the order of the key's Z

    def expected(self, r_mont, s_mont):
        """(Ar, Bs, Krs) affine as uint64 limb arrays (8,), (16,), (8,) for blinding r, s (Montgomery limbs)"""
        r = O._u64(r_mont).reshape(4); s = O._u64(s_mont).reshape(4)
        ar = _add(_add(self.k_alpha, self.dA), _mul(r, self.k_delta))
        bs = _add(_add(self.k_beta, self.dB), _mul(s, self.k_delta))
        krs = _add(_add(self.dK, self.dZ), _add(_mul(s, ar), _mul(r, bs)))
        krs = _sub(krs, _mul(_mul(r, s), self.k_delta))
        return (O.g1_from_scalars(ar.reshape(1, 4))[0], O.g2_from_scalars(bs.reshape(1, 4))[0], O.g1_from_scalars(krs.reshape(1, 4))[0])

    def check(self, proof256, r_mont, s_mont):
        p = np.ascontiguousarray(proof256, dtype=np.uint8).view(np.uint64)
        ar, bs, krs = self.expected(r_mont, s_mont)
        return bool(np.array_equal(p[0:8], ar) and np.array_equal(p[8:24], bs) and np.array_equal(p[24:32], krs))


def expected_commitment(seed, values_mont):
    """(commitment, knowledge proof) affine (8,) each for the committed values v over the synthetic Pedersen bases"""
    c = synth_dot(seed, G1_COMMIT_BASIS, values_mont)
    k = synth_dot(seed, G1_COMMIT_BASIS_SIGMA, values_mont)
    return O.g1_from_scalars(c.reshape(1, 4))[0], O.g1_from_scalars(k.reshape(1, 4))[0]
