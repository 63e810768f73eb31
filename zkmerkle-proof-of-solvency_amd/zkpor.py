"""ctypes binding of libzkpor.so (include/zkpor.h) — the Python-side stand-in for the cgo shim a maintainer adds to
the reference's src/prover.  There is NO fallback: if the HIP library is missing or no gfx950 device
is usable, construction raises."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZKPOR_LIB: experiment hook (kernel build variants under tools/bin/variants); the product library is the in-tree one
LIB_PATH = os.environ.get("ZKPOR_LIB") or os.path.join(_HERE, "libzkpor.so")

G1_A, G1_B, G1_K, G1_Z, G1_COMMIT_BASIS, G1_COMMIT_BASIS_SIGMA = range(6)
G2_B = 0
Z_ORDER_BITREV, Z_ORDER_NATURAL = 0, 1


class ZkporError(RuntimeError):
    pass


_lib = None


ABI_VERSION = 4   # include/zkpor.h ZKPOR_ABI_VERSION


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZkporError(f"{LIB_PATH} is not built (run __graft_entry__.build()); there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        lib.zkpor_abi_version.restype = ctypes.c_uint32
        got = int(lib.zkpor_abi_version())
        if got != ABI_VERSION:   # a stale binding would pass arguments in the old positions (include/zkpor.h ZKPOR_ABI_VERSION)
            raise ZkporError(f"{LIB_PATH} speaks ABI version {got}, this binding was written against {ABI_VERSION}")
        lib.zkpor_last_error.restype = ctypes.c_char_p
        lib.zkpor_phase_ms.restype = ctypes.c_double
        _lib = lib
    return _lib


def _p(a):
    if a is None:
        return None
    if isinstance(a, int):
        return ctypes.c_void_p(a)
    return a.ctypes.data_as(ctypes.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class DevBuf:
    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = nbytes
        p = ctypes.c_void_p()
        ctx._ck(ctx.lib.zkpor_dev_alloc(ctx.h, ctypes.c_size_t(nbytes), ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._ck(self.ctx.lib.zkpor_dev_upload(self.ctx.h, ctypes.c_void_p(self.ptr), _p(arr), ctypes.c_size_t(arr.nbytes)))
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._ck(self.ctx.lib.zkpor_dev_download(self.ctx.h, _p(out), ctypes.c_void_p(self.ptr), ctypes.c_size_t(out.nbytes)))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.zkpor_dev_free(self.ctx.h, ctypes.c_void_p(self.ptr))
            self.ptr = None


class Context:
    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.zkpor_init(ctypes.c_int(device), ctypes.c_void_p(stream) if stream else None, ctypes.byref(h))
        if rc != 0:
            raise ZkporError(f"zkpor_init failed with {rc} (no usable gfx950 device?)")
        self.h = h
        # ZKPOR_PARAMS="name=value,...": experiment hook (like ZKPOR_LIB) — parameters every context of the process starts with, so that a whole test file
        # can be run under a non-default setting
        for nv in filter(None, os.environ.get("ZKPOR_PARAMS", "").split(",")):
            name, _, val = nv.partition("=")
            self.set_param(name.strip(), int(val))

    def _ck(self, rc):
        if rc != 0:
            raise ZkporError(f"zkpor error {rc}: {self.lib.zkpor_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.lib.zkpor_destroy(self.h)
            self.h = None

    def set_param(self, name, value):
        self._ck(self.lib.zkpor_set_param(self.h, name.encode(), ctypes.c_int64(value)))

    def sync(self):
        self._ck(self.lib.zkpor_sync(self.h))

    def trim(self):
        """zkpor_trim: the context's grow-only scratch (workspace, staging area, NTT tables) back to the device; re-created on demand"""
        self._ck(self.lib.zkpor_trim(self.h))

    def phase_ms(self, name):
        calls = ctypes.c_uint64()
        ms = self.lib.zkpor_phase_ms(self.h, name.encode(), ctypes.byref(calls))
        return ms, calls.value

    def stat(self, name):
        v = ctypes.c_uint64()
        self._ck(self.lib.zkpor_stat(self.h, name.encode(), ctypes.byref(v)))
        return v.value

    def phase_reset(self):
        self.lib.zkpor_phase_reset(self.h)

    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    def fill_fr(self, buf, n, seed, kind=0):
        self._ck(self.lib.zkpor_dev_fill_fr(self.h, ctypes.c_void_p(buf.ptr), ctypes.c_size_t(n), ctypes.c_uint64(seed), ctypes.c_int(kind)))

    # ---- structured witness generation (include/zkpor.h zkpor_witgen_*) ----
    def witgen_poseidon_sboxes(self, t):
        self.lib.zkpor_witgen_poseidon_sboxes.restype = ctypes.c_size_t
        return int(self.lib.zkpor_witgen_poseidon_sboxes(ctypes.c_int(t)))

    def witgen_poseidon_trace_dev(self, t, d_states, count, d_trace):
        self._ck(self.lib.zkpor_witgen_poseidon_trace_dev(self.h, ctypes.c_int(t), ctypes.c_void_p(d_states), ctypes.c_size_t(count), ctypes.c_void_p(d_trace)))

    def witgen_poseidon_trace(self, states, t):
        """host convenience (tests): (final states, trace[(s * 3 + c), i]) of len(states) permutations of width t"""
        st = _u64(states).reshape(-1, t, 4)
        n = st.shape[0]
        ns = self.witgen_poseidon_sboxes(t)
        ds = self.alloc(st.nbytes).upload(st)
        dt = self.alloc(3 * ns * n * 32)
        try:
            self.witgen_poseidon_trace_dev(t, ds.ptr, n, dt.ptr)
            return ds.download(np.uint64, (n, t, 4)), dt.download(np.uint64, (3 * ns, n, 4))
        finally:
            ds.free(); dt.free()

    def witgen_limbs_dev(self, d_values, n, nb_limbs, d_limbs, d_mult, d_bad):
        self._ck(self.lib.zkpor_witgen_limbs_dev(self.h, ctypes.c_void_p(d_values), ctypes.c_size_t(n), ctypes.c_int(nb_limbs), ctypes.c_void_p(d_limbs),
                                                 ctypes.c_void_p(d_mult), ctypes.c_void_p(d_bad)))

    def witgen_limbs(self, values, nb_limbs):
        """host convenience (tests): (limbs[l, i] as Montgomery Fr, multiplicities of the 2^16 table, number of values out of range)"""
        v = _u64(values).reshape(-1, 4)
        n = v.shape[0]
        dv = self.alloc(v.nbytes).upload(v)
        dl = self.alloc(nb_limbs * n * 32)
        dm = self.alloc(65536 * 4).upload(np.zeros(65536, np.uint32))
        db = self.alloc(4).upload(np.zeros(1, np.uint32))
        try:
            self.witgen_limbs_dev(dv.ptr, n, nb_limbs, dl.ptr, dm.ptr, db.ptr)
            return dl.download(np.uint64, (nb_limbs, n, 4)), dm.download(np.uint32, (65536,)), int(db.download(np.uint32, (1,))[0])
        finally:
            for b in (dv, dl, dm, db):
                b.free()

    def witgen_inverse_dev(self, d_values, n, challenge, d_out, d_bad):
        c = _u64(challenge).reshape(4)
        self._ck(self.lib.zkpor_witgen_inverse_dev(self.h, ctypes.c_void_p(d_values), ctypes.c_size_t(n), _p(c), ctypes.c_void_p(d_out), ctypes.c_void_p(d_bad)))

    def witgen_inverse(self, values, challenge):
        v = _u64(values).reshape(-1, 4)
        n = v.shape[0]
        dv = self.alloc(v.nbytes).upload(v)
        do = self.alloc(v.nbytes)
        db = self.alloc(4).upload(np.zeros(1, np.uint32))
        try:
            self.witgen_inverse_dev(dv.ptr, n, challenge, do.ptr, db.ptr)
            return do.download(np.uint64, (n, 4)), int(db.download(np.uint32, (1,))[0])
        finally:
            for b in (dv, do, db):
                b.free()

    def witgen_bits_dev(self, d_values, n, nbits, d_bits, d_bad):
        self._ck(self.lib.zkpor_witgen_bits_dev(self.h, ctypes.c_void_p(d_values), ctypes.c_size_t(n), ctypes.c_int(nbits), ctypes.c_void_p(d_bits), ctypes.c_void_p(d_bad)))

    def witgen_bits(self, values, nbits):
        v = _u64(values).reshape(-1, 4)
        n = v.shape[0]
        dv = self.alloc(v.nbytes).upload(v); do = self.alloc(nbits * n * 32); db = self.alloc(4).upload(np.zeros(1, np.uint32))
        try:
            self.witgen_bits_dev(dv.ptr, n, nbits, do.ptr, db.ptr)
            return do.download(np.uint64, (nbits, n, 4)), int(db.download(np.uint32, (1,))[0])
        finally:
            for b in (dv, do, db):
                b.free()

    def witgen_gather_dev(self, d_table, table_len, d_indices, n, d_out, d_bad):
        self._ck(self.lib.zkpor_witgen_gather_dev(self.h, ctypes.c_void_p(d_table), ctypes.c_size_t(table_len), ctypes.c_void_p(d_indices), ctypes.c_size_t(n),
                                                  ctypes.c_void_p(d_out), ctypes.c_void_p(d_bad)))

    def witgen_gather(self, table, indices):
        t = _u64(table).reshape(-1, 4); ix = _u64(indices).reshape(-1, 4)
        n = ix.shape[0]
        dt = self.alloc(t.nbytes).upload(t); di = self.alloc(ix.nbytes).upload(ix); do = self.alloc(ix.nbytes); db = self.alloc(4).upload(np.zeros(1, np.uint32))
        try:
            self.witgen_gather_dev(dt.ptr, t.shape[0], di.ptr, n, do.ptr, db.ptr)
            return do.download(np.uint64, (n, 4)), int(db.download(np.uint32, (1,))[0])
        finally:
            for b in (dt, di, do, db):
                b.free()

    def witgen_divmod_small_dev(self, d_values, n, divisor, d_q, d_rem):
        self._ck(self.lib.zkpor_witgen_divmod_small_dev(self.h, ctypes.c_void_p(d_values), ctypes.c_size_t(n), ctypes.c_uint32(divisor), ctypes.c_void_p(d_q), ctypes.c_void_p(d_rem)))

    def witgen_divmod_small(self, values, divisor):
        v = _u64(values).reshape(-1, 4)
        n = v.shape[0]
        dv = self.alloc(v.nbytes).upload(v); dq = self.alloc(v.nbytes); dr = self.alloc(v.nbytes)
        try:
            self.witgen_divmod_small_dev(dv.ptr, n, divisor, dq.ptr, dr.ptr)
            return dq.download(np.uint64, (n, 4)), dr.download(np.uint64, (n, 4))
        finally:
            for b in (dv, dq, dr):
                b.free()

    def witgen_scatter_dev(self, d_w, d_src, d_wire_ids, n):
        self._ck(self.lib.zkpor_witgen_scatter_dev(self.h, ctypes.c_void_p(d_w), ctypes.c_void_p(d_src), ctypes.c_void_p(d_wire_ids), ctypes.c_size_t(n)))

    def witgen_scatter_known_dev(self, d_w, d_known, d_src, d_wire_ids, n):
        self._ck(self.lib.zkpor_witgen_scatter_known_dev(self.h, ctypes.c_void_p(d_w), ctypes.c_void_p(d_known), ctypes.c_void_p(d_src), ctypes.c_void_p(d_wire_ids), ctypes.c_size_t(n)))

    # ---- MSM ----
    def msm_g1(self, points, scalars):
        points = _u64(points); scalars = _u64(scalars)
        out = np.empty(12, dtype=np.uint64)
        self._ck(self.lib.zkpor_msm_g1(self.h, _p(points), _p(scalars), ctypes.c_size_t(scalars.reshape(-1, 4).shape[0]), _p(out)))
        return out

    def msm_g2(self, points, scalars):
        points = _u64(points); scalars = _u64(scalars)
        out = np.empty(24, dtype=np.uint64)
        self._ck(self.lib.zkpor_msm_g2(self.h, _p(points), _p(scalars), ctypes.c_size_t(scalars.reshape(-1, 4).shape[0]), _p(out)))
        return out

    def msm_g1_dev(self, d_points, d_scalars, n):
        out = np.empty(12, dtype=np.uint64)
        self._ck(self.lib.zkpor_msm_g1_dev(self.h, ctypes.c_void_p(d_points), ctypes.c_void_p(d_scalars), ctypes.c_size_t(n), _p(out)))
        return out

    def msm_g2_dev(self, d_points, d_scalars, n):
        out = np.empty(24, dtype=np.uint64)
        self._ck(self.lib.zkpor_msm_g2_dev(self.h, ctypes.c_void_p(d_points), ctypes.c_void_p(d_scalars), ctypes.c_size_t(n), _p(out)))
        return out

    def msm_digits_dev(self, d_scalars, n, tables=1, absent0=None, absent1=None, cap=None):
        """test-facing (zkpor_msm_digits_dev): the signed-digit stream of n resident scalars grouped by bucket -> (keys, vals, info dict)"""
        info = np.zeros(8, dtype=np.uint64)
        a0 = None if absent0 is None else np.ascontiguousarray(absent0, dtype=np.uint8)
        a1 = None if absent1 is None else np.ascontiguousarray(absent1, dtype=np.uint8)
        if cap is None:
            cap = n * 130
        keys = np.empty(cap, dtype=np.uint32); vals = np.empty(cap, dtype=np.uint32)
        self._ck(self.lib.zkpor_msm_digits_dev(self.h, ctypes.c_void_p(d_scalars), ctypes.c_size_t(n), ctypes.c_int(tables), _p(a0) if a0 is not None else None,
                                                _p(a1) if a1 is not None else None, _p(keys), _p(vals), ctypes.c_size_t(cap), _p(info)))
        m = int(info[0])
        names = ("entries", "entries_group0", "entries_group1", "c", "W", "piece", "bpw", "levels")
        return keys[:m], vals[:m], {k: int(v) for k, v in zip(names, info)}

    # ---- NTT / H ----
    def fft(self, a, log2n, inverse=False, decimation=1, on_coset=False):
        a = _u64(a).copy()
        self._ck(self.lib.zkpor_fft(self.h, _p(a), ctypes.c_int(log2n), ctypes.c_int(int(inverse)), ctypes.c_int(decimation), ctypes.c_int(int(on_coset))))
        return a

    def fft_dev(self, d_a, log2n, inverse=False, decimation=1, on_coset=False):
        self._ck(self.lib.zkpor_fft_dev(self.h, ctypes.c_void_p(d_a), ctypes.c_int(log2n), ctypes.c_int(int(inverse)), ctypes.c_int(decimation), ctypes.c_int(int(on_coset))))

    def compute_h(self, a, b, c, log2_domain):
        a = _u64(a); b = _u64(b); c = _u64(c)
        out = np.empty((1 << log2_domain, 4), dtype=np.uint64)
        self._ck(self.lib.zkpor_compute_h(self.h, ctypes.c_int(log2_domain), _p(a), _p(b), _p(c), ctypes.c_size_t(a.shape[0]), _p(out)))
        return out

    def compute_h_dev(self, log2_domain, d_a, d_b, d_c):
        self._ck(self.lib.zkpor_compute_h_dev(self.h, ctypes.c_int(log2_domain), ctypes.c_void_p(d_a), ctypes.c_void_p(d_b), ctypes.c_void_p(d_c)))

    def compute_h_shard_dev(self, log2_domain, world_log2, rank, d_a, d_b, d_c, step):
        vp = lambda x: ctypes.c_void_p(x) if x else None
        self._ck(self.lib.zkpor_compute_h_shard_dev(self.h, ctypes.c_int(log2_domain), ctypes.c_int(world_log2), ctypes.c_int(rank),
                                                     vp(d_a), vp(d_b), vp(d_c), ctypes.c_int(step)))

    def shard_transpose_dev(self, d_out, d_in, log2_local, world_log2, interleave):
        self._ck(self.lib.zkpor_shard_transpose_dev(self.h, ctypes.c_void_p(d_out), ctypes.c_void_p(d_in), ctypes.c_int(log2_local),
                                                     ctypes.c_int(world_log2), ctypes.c_int(int(interleave))))

    # ---- Groth16 ----
    def prove_tail(self, pk, w, a, b, c, r, s):
        w = _u64(w); a = _u64(a); b = _u64(b); c = _u64(c); r = _u64(r); s = _u64(s)
        out = np.empty(256, dtype=np.uint8)
        self._ck(self.lib.zkpor_prove_tail(self.h, pk.h, _p(w), _p(a), _p(b), _p(c), ctypes.c_size_t(a.shape[0]), _p(r), _p(s), _p(out)))
        return out

    def prove_r1cs(self, pk, r1cs, w, r, s):
        """host-pointer form with the constraint matrices resident: only w crosses PCIe (zkpor_prove_r1cs)"""
        w = _u64(w); r = _u64(r); s = _u64(s)
        out = np.empty(256, dtype=np.uint8)
        self._ck(self.lib.zkpor_prove_r1cs(self.h, pk.h, r1cs.h, _p(w), _p(r), _p(s), _p(out)))
        return out

    def prove_inputs(self, pk, r1cs, solver, inputs, r, s):
        """groth16.Prove from the assigned inputs: solver program, a / b / c and the prove tail on the device (zkpor_prove_inputs)"""
        inputs = _u64(inputs).reshape(-1, 4); r = _u64(r); s = _u64(s)
        out = np.empty(256, dtype=np.uint8)
        self._ck(self.lib.zkpor_prove_inputs(self.h, pk.h, r1cs.h, solver.h, _p(inputs), ctypes.c_size_t(inputs.shape[0]), _p(r), _p(s), _p(out)))
        return out

    def prove_tail_dev(self, pk, d_w, d_a, d_b, d_c, r, s):
        r = _u64(r); s = _u64(s)
        out = np.empty(256, dtype=np.uint8)
        self._ck(self.lib.zkpor_prove_tail_dev(self.h, pk.h, ctypes.c_void_p(d_w), ctypes.c_void_p(d_a), ctypes.c_void_p(d_b), ctypes.c_void_p(d_c), _p(r), _p(s), _p(out)))
        return out

    def prove_tail_dev_keep(self, pk, d_w, d_a, d_b, d_c, d_wa, d_wb, d_wc, r, s):
        """zkpor_prove_tail_dev_keep: a, b, c are only read; the work buffers are overwritten (h is left in d_wa)"""
        r = _u64(r); s = _u64(s)
        out = np.empty(256, dtype=np.uint8)
        vp = ctypes.c_void_p
        self._ck(self.lib.zkpor_prove_tail_dev_keep(self.h, pk.h, vp(d_w), vp(d_a), vp(d_b), vp(d_c), vp(d_wa), vp(d_wb), vp(d_wc), _p(r), _p(s), _p(out)))
        return out

    def prove_sums_dev(self, pk, d_w, d_h):
        """the five multi-exponentiations over the key's (or shard's) arrays: 576 B of Jacobian points A.w | B1.w | B2.w | K.w | Z.h"""
        out = np.empty(576, dtype=np.uint8)
        self._ck(self.lib.zkpor_prove_sums_dev(self.h, pk.h, ctypes.c_void_p(d_w), ctypes.c_void_p(d_h) if d_h else None, _p(out)))
        return out

    def commit(self, pk, values):
        values = _u64(values)
        c = np.empty(8, dtype=np.uint64); k = np.empty(8, dtype=np.uint64)
        self._ck(self.lib.zkpor_commit(self.h, pk.h, _p(values), ctypes.c_size_t(values.reshape(-1, 4).shape[0]), _p(c), _p(k)))
        return c, k


def proof_write_raw(proof256, commitments=None, pok=None):
    lib = load_library()
    proof256 = np.ascontiguousarray(proof256, dtype=np.uint8)
    ncom = 0 if commitments is None else np.asarray(commitments).reshape(-1, 8).shape[0]
    com = None if commitments is None else _u64(commitments)
    pk_ = None if pok is None else _u64(pok)
    out = np.empty(256 + 4 + 64 * ncom + 64, dtype=np.uint8)
    n = ctypes.c_size_t()
    rc = lib.zkpor_proof_write_raw(_p(proof256), _p(com), ctypes.c_uint32(ncom), _p(pk_), _p(out), ctypes.c_size_t(out.size), ctypes.byref(n))
    if rc != 0:
        raise ZkporError(f"zkpor_proof_write_raw failed: {rc}")
    return out[: n.value]


def prove_assemble(consts, sums, r, s):
    """host only: blinding + assembly of the proof (256 B, as prove_tail returns it) from the five added sums (576 B)"""
    lib = load_library()
    alpha, beta, delta, beta2, delta2 = (_u64(x) for x in consts)
    sums = np.ascontiguousarray(sums, dtype=np.uint8).reshape(576)
    r = _u64(r); s = _u64(s)
    out = np.empty(256, dtype=np.uint8)
    rc = lib.zkpor_prove_assemble(_p(alpha), _p(beta), _p(delta), _p(beta2), _p(delta2), _p(sums), _p(r), _p(s), _p(out))
    if rc != 0:
        raise ZkporError(f"zkpor_prove_assemble failed: {rc}")
    return out


def g1_jac_sum(parts):
    """host-side sum of Jacobian G1 points (n,12) uint64 -> (12,) — combines partial MSM results, needs no GPU"""
    lib = load_library()
    parts = _u64(parts).reshape(-1, 12)
    out = np.empty(12, dtype=np.uint64)
    rc = lib.zkpor_g1_jac_sum(_p(parts), ctypes.c_size_t(parts.shape[0]), _p(out))
    if rc != 0:
        raise ZkporError(f"zkpor_g1_jac_sum failed: {rc}")
    return out


def g2_jac_sum(parts):
    lib = load_library()
    parts = _u64(parts).reshape(-1, 24)
    out = np.empty(24, dtype=np.uint64)
    rc = lib.zkpor_g2_jac_sum(_p(parts), ctypes.c_size_t(parts.shape[0]), _p(out))
    if rc != 0:
        raise ZkporError(f"zkpor_g2_jac_sum failed: {rc}")
    return out


def msm_split_g1(ctx, d_points, d_scalars, n, rank, world, dist=None):
    """config 5 of BASELINE.json: one multi-exponentiation sharded by contiguous index range over `world` GPUs.
    Every rank sums its slice on its own device; the `world` partial Jacobian points (96 B each) are all-gathered
    (the only collective on the path, latency-bound) and added on the host.  With dist=None the ranks are emulated
    sequentially on one device (parity test)."""
    lo = n * rank // world
    hi = n * (rank + 1) // world
    part = ctx.msm_g1_dev(d_points + 64 * lo, d_scalars + 32 * lo, hi - lo)
    if dist is None:
        return part
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.from_numpy(part.view(np.int64).copy()).to(dev)
    allp = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allp, mine)
    parts = np.stack([t.cpu().numpy().view(np.uint64) for t in allp])
    return g1_jac_sum(parts)


class PkLayout(ctypes.Structure):
    """zkpor_pk_layout_t (include/zkpor.h): counts and byte offsets of the sections of a gnark pk.WriteTo stream"""
    _fields_ = [(n, ctypes.c_uint64) for n in (
        "domain_cardinality", "n_a", "n_b1", "n_z", "n_k", "n_b2", "off_alpha", "off_a", "off_b1", "off_z", "off_k", "off_beta2",
        "off_b2", "n_wires", "n_inf_a", "n_inf_b", "off_inf_a", "off_inf_b", "n_basis", "off_basis", "n_basis_sigma",
        "off_basis_sigma", "bytes_total")] + [("domain_header_bytes", ctypes.c_uint32), ("n_commitment_keys", ctypes.c_uint32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def pk_gnark_layout(data):
    """walk the headers of a gnark proving-key stream on the host (no device): dict of counts/offsets, ZkporError if malformed"""
    lib = load_library()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    out = PkLayout(); err = ctypes.create_string_buffer(512)
    rc = lib.zkpor_pk_gnark_layout(_p(buf), ctypes.c_size_t(buf.size), ctypes.byref(out), err, ctypes.c_size_t(512))
    if rc != 0:
        raise ZkporError(f"zkpor error {rc}: {err.value.decode()}")
    return out.as_dict()


class ProvingKey:
    """HBM-resident proving key (the device half of gnark's groth16.ProvingKey)"""

    def __init__(self, ctx):
        self.ctx = ctx
        h = ctypes.c_void_p()
        ctx._ck(ctx.lib.zkpor_pk_create(ctx.h, ctypes.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.ctx.lib.zkpor_pk_destroy(self.h)
            self.h = None

    def set_g1(self, which, pts):
        pts = _u64(pts).reshape(-1, 8)
        self.ctx._ck(self.ctx.lib.zkpor_pk_set_g1(self.h, ctypes.c_int(which), _p(pts), ctypes.c_size_t(pts.shape[0])))

    def set_g2(self, which, pts):
        pts = _u64(pts).reshape(-1, 16)
        self.ctx._ck(self.ctx.lib.zkpor_pk_set_g2(self.h, ctypes.c_int(which), _p(pts), ctypes.c_size_t(pts.shape[0])))

    def set_g1_compressed(self, which, comp32):
        """gnark-crypto compressed G1 points (n x 32 B, as pk.WriteTo stores them); decompressed on the device"""
        comp32 = np.ascontiguousarray(comp32, dtype=np.uint8).reshape(-1, 32)
        self.ctx._ck(self.ctx.lib.zkpor_pk_set_g1_compressed(self.h, ctypes.c_int(which), _p(comp32), ctypes.c_size_t(comp32.shape[0])))

    def set_g2_compressed(self, which, comp64):
        comp64 = np.ascontiguousarray(comp64, dtype=np.uint8).reshape(-1, 64)
        self.ctx._ck(self.ctx.lib.zkpor_pk_set_g2_compressed(self.h, ctypes.c_int(which), _p(comp64), ctypes.c_size_t(comp64.shape[0])))

    def set_consts(self, alpha, beta, delta, beta2, delta2, log2_domain, inf_a, inf_b, n_wires, n_public, committed_idx=None, z_order=Z_ORDER_BITREV):
        ia = None if inf_a is None else np.ascontiguousarray(inf_a, dtype=np.uint8)
        ib = None if inf_b is None else np.ascontiguousarray(inf_b, dtype=np.uint8)
        ci = None if committed_idx is None else np.ascontiguousarray(committed_idx, dtype=np.uint32)
        self.ctx._ck(self.ctx.lib.zkpor_pk_set_consts(
            self.h, _p(_u64(alpha)), _p(_u64(beta)), _p(_u64(delta)), _p(_u64(beta2)), _p(_u64(delta2)), ctypes.c_int(log2_domain),
            _p(ia), _p(ib), ctypes.c_size_t(n_wires), ctypes.c_size_t(n_public), _p(ci), ctypes.c_size_t(0 if ci is None else ci.size),
            ctypes.c_int(z_order)))

    def load_gnark(self, src, n_public, committed_idx=None, z_order=Z_ORDER_BITREV):
        """load a whole key from gnark's pk.WriteTo container (src/keygen/main.go:46): `src` is a path or the bytes;
        n_public counts the ONE wire; committed_idx = committed + commitment wire indices (r1cs.CommitmentInfo).
        Returns the container layout as a dict."""
        ci = None if committed_idx is None else np.ascontiguousarray(committed_idx, dtype=np.uint32)
        nci = ctypes.c_size_t(0 if ci is None else ci.size)
        info = PkLayout()
        if isinstance(src, (str, os.PathLike)):
            rc = self.ctx.lib.zkpor_pk_load_gnark(self.h, os.fsencode(src), ctypes.c_size_t(n_public), _p(ci), nci, ctypes.c_int(z_order), ctypes.byref(info))
        else:
            buf = np.frombuffer(bytes(src), dtype=np.uint8)
            rc = self.ctx.lib.zkpor_pk_load_gnark_mem(self.h, _p(buf), ctypes.c_size_t(buf.size), ctypes.c_size_t(n_public), _p(ci), nci, ctypes.c_int(z_order), ctypes.byref(info))
        self.ctx._ck(rc)
        return info.as_dict()

    def load_gnark_shard(self, src, n_public, wire_lo, wire_hi, z_lo, z_hi, committed_idx=None, z_order=Z_ORDER_BITREV):
        """one rank's share of a split key straight from the container (path or bytes): see zkpor_pk_load_gnark_shard"""
        ci = None if committed_idx is None else np.ascontiguousarray(committed_idx, dtype=np.uint32)
        nci = ctypes.c_size_t(0 if ci is None else ci.size)
        info = PkLayout()
        rng = [ctypes.c_size_t(v) for v in (wire_lo, wire_hi, z_lo, z_hi)]
        if isinstance(src, (str, os.PathLike)):
            rc = self.ctx.lib.zkpor_pk_load_gnark_shard(self.h, os.fsencode(src), ctypes.c_size_t(n_public), _p(ci), nci, *rng, ctypes.c_int(z_order), ctypes.byref(info))
        else:
            buf = np.frombuffer(bytes(src), dtype=np.uint8)
            rc = self.ctx.lib.zkpor_pk_load_gnark_shard_mem(self.h, _p(buf), ctypes.c_size_t(buf.size), ctypes.c_size_t(n_public), _p(ci), nci, *rng, ctypes.c_int(z_order), ctypes.byref(info))
        self.ctx._ck(rc)
        return info.as_dict()

    def keep_range(self, wire_lo, wire_hi, z_lo, z_hi):
        """turn the loaded key into a shard of the single-proof split (zkpor_pk_keep_range)"""
        self.ctx._ck(self.ctx.lib.zkpor_pk_keep_range(self.h, ctypes.c_size_t(wire_lo), ctypes.c_size_t(wire_hi), ctypes.c_size_t(z_lo), ctypes.c_size_t(z_hi)))

    def consts(self):
        """(alpha, beta, delta, beta2, delta2) of the loaded key as uint64 limb arrays"""
        a = np.empty(8, np.uint64); b = np.empty(8, np.uint64); d = np.empty(8, np.uint64)
        b2 = np.empty(16, np.uint64); d2 = np.empty(16, np.uint64)
        self.ctx._ck(self.ctx.lib.zkpor_pk_consts(self.h, _p(a), _p(b), _p(d), _p(b2), _p(d2)))
        return a, b, d, b2, d2

    def synth(self, log2_domain, n_wires, n_public, n_committed, seed):
        self.ctx._ck(self.ctx.lib.zkpor_pk_synth(self.h, ctypes.c_int(log2_domain), ctypes.c_size_t(n_wires), ctypes.c_size_t(n_public),
                                                 ctypes.c_size_t(n_committed), ctypes.c_uint64(seed)))

    def synth_masked(self, log2_domain, n_wires, n_public, inf_a, inf_b, removed_idx, n_basis, seed):
        """the synthetic key with a circuit's sparsity: A / B infinity masks (bytes per wire), the wires K leaves out besides the public ones"""
        inf_a = np.ascontiguousarray(inf_a, dtype=np.uint8); inf_b = np.ascontiguousarray(inf_b, dtype=np.uint8)
        removed_idx = np.ascontiguousarray(removed_idx, dtype=np.uint32)
        assert inf_a.shape[0] == n_wires and inf_b.shape[0] == n_wires
        self.ctx._ck(self.ctx.lib.zkpor_pk_synth_masked(self.h, ctypes.c_int(log2_domain), ctypes.c_size_t(n_wires), ctypes.c_size_t(n_public), _p(inf_a), _p(inf_b),
                                                        _p(removed_idx), ctypes.c_size_t(removed_idx.shape[0]), ctypes.c_size_t(n_basis), ctypes.c_uint64(seed)))

    def g1_dev(self, which):
        p = ctypes.c_void_p(); n = ctypes.c_size_t()
        self.ctx._ck(self.ctx.lib.zkpor_pk_g1_dev(self.h, ctypes.c_int(which), ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def g2_dev(self, which=G2_B):
        p = ctypes.c_void_p(); n = ctypes.c_size_t()
        self.ctx._ck(self.ctx.lib.zkpor_pk_g2_dev(self.h, ctypes.c_int(which), ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value


# ---- the trapdoor of ProvingKey.synth (mirrors csrc/groth16.hip synth_k / synth_is_inf; used by parity tests) ----
_M64 = (1 << 64) - 1
SYNTH_RUN = 32
SYNTH_Q = 0x9e3779b97f4a7c15


def _smix(x):
    x = (x + 0x9e3779b97f4a7c15) & _M64
    x = ((x ^ (x >> 30)) * 0xbf58476d1ce4e5b9) & _M64
    x = ((x ^ (x >> 27)) * 0x94d049bb133111eb) & _M64
    return x ^ (x >> 31)


def synth_scalar(seed, arr, i):
    """integer s with P_i = s*G for point i of synthetic array `arr` (ignoring the infinity pattern)"""
    run, j = divmod(i, SYNTH_RUN)
    x = seed ^ (((arr + 1) * 0xa0761d6478bd642f) & _M64) ^ ((run * 0xe7037ed1a0b428db) & _M64)
    return (_smix(x) | 1) + j * SYNTH_Q


def synth_is_inf(i, mod):
    if not mod:
        return False
    x = (i * 0xd6e8feb86659fd93) & _M64
    x ^= x >> 32
    return x % mod == 0


SYNTH_INF_MOD = {G1_A: 64, G1_B: 10, G1_K: 4, G1_Z: 0, G1_COMMIT_BASIS: 0, G1_COMMIT_BASIS_SIGMA: 0}


# ---- Poseidon account tree (methods attached to Context) ----
ACCOUNT_DTYPE = np.dtype([("id_be", np.uint8, 32), ("equity", np.uint64, 2), ("debt", np.uint64, 2),
                          ("collateral", np.uint64, 2), ("n_assets", np.uint32), ("asset_off", np.uint32)])
ASSET_DTYPE = np.dtype([("equity", np.uint64), ("debt", np.uint64), ("loan", np.uint64), ("margin", np.uint64),
                        ("portfolio_margin", np.uint64), ("index", np.uint32), ("pad", np.uint32)])


def _poseidon_hash(self, inputs, length):
    """`count` independent poseidon.Poseidon(inputs...) of `length` Montgomery Fr each"""
    inputs = _u64(inputs).reshape(-1, 4)
    count = inputs.shape[0] // length
    out = np.empty((count, 4), dtype=np.uint64)
    self._ck(self.lib.zkpor_poseidon_hash(self.h, _p(inputs), ctypes.c_size_t(length), ctypes.c_size_t(count), _p(out)))
    return out


def _poseidon_leaves(self, accounts, assets, tier):
    accounts = np.ascontiguousarray(accounts, dtype=ACCOUNT_DTYPE)
    assets = np.ascontiguousarray(assets, dtype=ASSET_DTYPE)
    out = np.empty((accounts.shape[0], 32), dtype=np.uint8)
    self._ck(self.lib.zkpor_poseidon_leaves(self.h, _p(accounts), _p(assets), ctypes.c_size_t(assets.shape[0]),
                                            ctypes.c_size_t(accounts.shape[0]), ctypes.c_int(tier), _p(out)))
    return out


def _merkle_build(self, leaves_be, depth, nil_leaf_be, want_levels=False):
    leaves_be = np.ascontiguousarray(leaves_be, dtype=np.uint8).reshape(-1, 32)
    n = leaves_be.shape[0]
    nil_leaf_be = np.ascontiguousarray(nil_leaf_be, dtype=np.uint8)
    tot = sum((n + (1 << l) - 1) >> l for l in range(1, depth + 1))
    levels = np.empty((tot, 32), dtype=np.uint8) if want_levels else None
    root = np.empty(32, dtype=np.uint8)
    self._ck(self.lib.zkpor_merkle_build(self.h, _p(leaves_be), ctypes.c_size_t(n), ctypes.c_int(depth), _p(nil_leaf_be), _p(levels), _p(root)))
    return root, levels


def _merkle_build_dev(self, d_leaves, n, depth, nil_leaf_mont):
    nil_leaf_mont = _u64(nil_leaf_mont)
    root = np.empty(4, dtype=np.uint64)
    self._ck(self.lib.zkpor_merkle_build_dev(self.h, ctypes.c_void_p(d_leaves), ctypes.c_size_t(n), ctypes.c_int(depth), _p(nil_leaf_mont), _p(root)))
    return root


def _g1_decompress(self, comp32):
    comp32 = np.ascontiguousarray(comp32, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros((comp32.shape[0], 8), dtype=np.uint64)
    self._ck(self.lib.zkpor_g1_decompress(self.h, _p(comp32), ctypes.c_size_t(comp32.shape[0]), _p(out)))
    return out


def _g2_decompress(self, comp64):
    comp64 = np.ascontiguousarray(comp64, dtype=np.uint8).reshape(-1, 64)
    out = np.zeros((comp64.shape[0], 16), dtype=np.uint64)
    self._ck(self.lib.zkpor_g2_decompress(self.h, _p(comp64), ctypes.c_size_t(comp64.shape[0]), _p(out)))
    return out


TIER_DTYPE = np.dtype([("boundary", np.uint64, 2), ("ratio", np.uint8), ("pad", np.uint8, 7)])
CEX_CONST_DTYPE = np.dtype([("base_price", np.uint64), ("loan", TIER_DTYPE, 12), ("margin", TIER_DTYPE, 12), ("portfolio_margin", TIER_DTYPE, 12)])
CEX_TOTALS_DTYPE = np.dtype([("total_equity", np.uint64), ("total_debt", np.uint64), ("loan_collateral", np.uint64),
                             ("margin_collateral", np.uint64), ("portfolio_margin_collateral", np.uint64)])


def _cex_commitments(self, consts, totals):
    """one 32-byte commitment per CEX state: totals[n_states, n_assets] rows over the constant asset table"""
    consts = np.ascontiguousarray(consts, dtype=CEX_CONST_DTYPE); totals = np.ascontiguousarray(totals, dtype=CEX_TOTALS_DTYPE)
    n_assets = consts.shape[0]; n_states = totals.size // n_assets
    out = np.empty((n_states, 32), dtype=np.uint8)
    self._ck(self.lib.zkpor_cex_commitments(self.h, _p(consts), ctypes.c_size_t(n_assets), _p(totals), ctypes.c_size_t(n_states), _p(out)))
    return out


def _batch_commitments(self, roots, before, after, min_idx, max_idx):
    roots = np.ascontiguousarray(roots, dtype=np.uint8).reshape(-1, 32); n = roots.shape[0]
    before = np.ascontiguousarray(before, dtype=np.uint8).reshape(n, 32); after = np.ascontiguousarray(after, dtype=np.uint8).reshape(n, 32)
    min_idx = np.ascontiguousarray(min_idx, dtype=np.uint32); max_idx = np.ascontiguousarray(max_idx, dtype=np.uint32)
    out = np.empty((n, 32), dtype=np.uint8)
    self._ck(self.lib.zkpor_batch_commitments(self.h, _p(roots), _p(before), _p(after), _p(min_idx), _p(max_idx), ctypes.c_size_t(n), _p(out)))
    return out


def _account_totals(self, accounts, assets, consts, want_tiers=False):
    """(accounts with equity/debt/collateral filled, valid[n], tier_info[n_assets_total, 6] or None)"""
    accounts = np.ascontiguousarray(accounts, dtype=ACCOUNT_DTYPE).copy(); assets = np.ascontiguousarray(assets, dtype=ASSET_DTYPE)
    consts = np.ascontiguousarray(consts, dtype=CEX_CONST_DTYPE)
    valid = np.zeros(accounts.shape[0], dtype=np.uint8)
    tiers = np.zeros((assets.shape[0], 6), dtype=np.uint8) if want_tiers else None
    self._ck(self.lib.zkpor_account_totals(self.h, _p(accounts), _p(assets), ctypes.c_size_t(assets.shape[0]), ctypes.c_size_t(accounts.shape[0]),
                                           _p(consts), ctypes.c_size_t(consts.shape[0]), _p(tiers), _p(valid)))
    return accounts, valid, tiers


Context.account_totals = _account_totals
Context.cex_commitments = _cex_commitments
Context.batch_commitments = _batch_commitments
Context.g1_decompress = _g1_decompress
Context.g2_decompress = _g2_decompress
Context.poseidon_hash = _poseidon_hash
Context.poseidon_leaves = _poseidon_leaves
Context.merkle_build = _merkle_build
Context.merkle_build_dev = _merkle_build_dev


class FixedDepthMerkleTree:
    """Mirror of the reference's merkletree.FixedDepthMerkleTree (src/utils/merkletree/merkletree.go:27-52) over the
    device-resident tree of libzkpor: same method names, same two-phase Set -> Build -> GetProof/Root usage, hashes as
    32-byte big-endian `bytes`.  The hasher is fixed to poseidon.NewPoseidon (account_tree.go:14-23).  Where the
    reference panics (constructor) or returns an error (Set, GetProof) this raises ZkporError."""

    def __init__(self, ctx, depth, nil_leaf_hash, capacity):
        self.ctx = ctx
        self.depth = depth
        self.capacity = capacity
        nil = np.frombuffer(bytes(nil_leaf_hash), dtype=np.uint8)
        if nil.shape[0] != 32:
            raise ZkporError("nil leaf hash must be 32 bytes")
        h = ctypes.c_void_p()
        ctx._ck(ctx.lib.zkpor_tree_create(ctx.h, ctypes.c_int(depth), _p(nil), ctypes.c_uint64(capacity), ctypes.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.ctx.lib.zkpor_tree_destroy(self.h)
            self.h = None

    def nil_hash(self, level):
        out = np.empty(32, dtype=np.uint8)
        self.ctx._ck(self.ctx.lib.zkpor_tree_nil_hash(self.h, ctypes.c_int(level), _p(out)))
        return out.tobytes()

    def set(self, key, value):
        self.set_many([key], np.frombuffer(bytes(value), dtype=np.uint8))

    def set_many(self, keys, values_be):
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        values_be = np.ascontiguousarray(values_be, dtype=np.uint8).reshape(-1, 32)
        assert values_be.shape[0] == keys.shape[0]
        self.ctx._ck(self.ctx.lib.zkpor_tree_set(self.h, _p(keys), _p(values_be), ctypes.c_size_t(keys.shape[0])))

    def set_range_dev(self, first_key, d_leaves_mont, n):
        self.ctx._ck(self.ctx.lib.zkpor_tree_set_range_dev(self.h, ctypes.c_uint64(first_key), ctypes.c_void_p(d_leaves_mont), ctypes.c_size_t(n)))

    def set_accounts(self, first_key, accounts, assets, tier, consts=None):
        """leaves of a chunk of accounts computed and Set on the device; returns (accounts with totals, valid) when the CEX
        table is given (totals computed on the device too), else None"""
        accounts = np.ascontiguousarray(accounts, dtype=ACCOUNT_DTYPE).copy(); assets = np.ascontiguousarray(assets, dtype=ASSET_DTYPE)
        valid = np.zeros(accounts.shape[0], dtype=np.uint8) if consts is not None else None
        c = np.ascontiguousarray(consts, dtype=CEX_CONST_DTYPE) if consts is not None else None
        self.ctx._ck(self.ctx.lib.zkpor_tree_set_accounts(self.h, ctypes.c_uint64(first_key), _p(accounts), _p(assets), ctypes.c_size_t(assets.shape[0]),
                                                        ctypes.c_size_t(accounts.shape[0]), ctypes.c_int(tier), _p(c),
                                                        ctypes.c_size_t(0 if c is None else c.shape[0]), _p(valid)))
        return (accounts, valid) if consts is not None else None

    def build(self):
        self.ctx._ck(self.ctx.lib.zkpor_tree_build(self.h))

    def root(self):
        out = np.empty(32, dtype=np.uint8)
        self.ctx._ck(self.ctx.lib.zkpor_tree_root(self.h, _p(out)))
        return out.tobytes()

    def get_many(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        out = np.empty((keys.shape[0], 32), dtype=np.uint8)
        self.ctx._ck(self.ctx.lib.zkpor_tree_get(self.h, _p(keys), ctypes.c_size_t(keys.shape[0]), _p(out)))
        return out

    def get(self, key):
        return self.get_many([key])[0].tobytes()

    def get_proofs(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        out = np.empty((keys.shape[0], self.depth, 32), dtype=np.uint8)
        self.ctx._ck(self.ctx.lib.zkpor_tree_get_proofs(self.h, _p(keys), ctypes.c_size_t(keys.shape[0]), _p(out)))
        return out

    def get_proof(self, key):
        return [p.tobytes() for p in self.get_proofs([key])[0]]


def verify_proofs(ctx, root, keys, proofs, leaves_be, depth):
    """merkletree.VerifyProof (merkletree.go:334-355) for a batch; returns a bool array"""
    root = np.frombuffer(bytes(root), dtype=np.uint8)
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    proofs = np.ascontiguousarray(proofs, dtype=np.uint8).reshape(keys.shape[0], -1)
    if proofs.shape[1] != depth * 32:
        return np.zeros(keys.shape[0], dtype=bool)  # len(proof) != depth
    leaves_be = np.ascontiguousarray(leaves_be, dtype=np.uint8).reshape(-1, 32)
    ok = np.zeros(keys.shape[0], dtype=np.uint8)
    ctx._ck(ctx.lib.zkpor_merkle_verify_proofs(ctx.h, _p(root), _p(keys), _p(proofs), _p(leaves_be),
                                              ctypes.c_size_t(keys.shape[0]), ctypes.c_int(depth), _p(ok)))
    return ok.astype(bool)


def verify_proof(ctx, root, key, proof, leaf, depth):
    if len(proof) != depth or key >= (1 << depth):
        return False
    return bool(verify_proofs(ctx, root, [key], np.frombuffer(b"".join(proof), dtype=np.uint8), np.frombuffer(bytes(leaf), dtype=np.uint8), depth)[0])


class R1CS:
    """constraint matrices resident on the device (include/zkpor.h zkpor_r1cs_*): a, b, c = L.w, R.w, O.w"""

    def __init__(self, ctx, n_constraints, n_wires, coeff_table):
        self.ctx = ctx
        self.n_constraints = n_constraints; self.n_wires = n_wires
        coeff_table = _u64(coeff_table).reshape(-1, 4)
        h = ctypes.c_void_p()
        ctx._ck(ctx.lib.zkpor_r1cs_create(ctx.h, ctypes.c_size_t(n_constraints), ctypes.c_size_t(n_wires), _p(coeff_table),
                                          ctypes.c_size_t(coeff_table.shape[0]), ctypes.byref(h)))
        self.h = h

    def set_matrix(self, which, row_ptr, coeff_ids, wire_ids):
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        coeff_ids = np.ascontiguousarray(coeff_ids, dtype=np.uint32); wire_ids = np.ascontiguousarray(wire_ids, dtype=np.uint32)
        assert row_ptr.shape[0] == self.n_constraints + 1 and coeff_ids.shape == wire_ids.shape
        self.ctx._ck(self.ctx.lib.zkpor_r1cs_set_matrix(self.h, ctypes.c_int(which), _p(row_ptr), _p(coeff_ids), _p(wire_ids), ctypes.c_size_t(coeff_ids.shape[0])))

    def eval(self, w):
        w = _u64(w).reshape(-1, 4)
        assert w.shape[0] == self.n_wires
        a = np.empty((self.n_constraints, 4), np.uint64); b = np.empty_like(a); c = np.empty_like(a)
        self.ctx._ck(self.ctx.lib.zkpor_r1cs_eval(self.h, _p(w), _p(a), _p(b), _p(c)))
        return a, b, c

    def check_dev(self, d_w):
        """(number of constraints that do not hold for the wire vector on the device, lowest failing row or None)"""
        c = (ctypes.c_uint64 * 2)()
        self.ctx._ck(self.ctx.lib.zkpor_r1cs_check_dev(self.h, ctypes.c_void_p(d_w), c))
        return int(c[0]), (int(c[1]) if c[0] else None)

    def eval_dev(self, d_w, d_a, d_b, d_c, domain_size, ctx=None):
        """ctx: queue on another context of the same GPU (a second worker); default the R1CS's own"""
        if ctx is None:
            self.ctx._ck(self.ctx.lib.zkpor_r1cs_eval_dev(self.h, ctypes.c_void_p(d_w), ctypes.c_void_p(d_a), ctypes.c_void_p(d_b), ctypes.c_void_p(d_c), ctypes.c_size_t(domain_size)))
        else:
            ctx._ck(ctx.lib.zkpor_r1cs_eval_on(ctx.h, self.h, ctypes.c_void_p(d_w), ctypes.c_void_p(d_a), ctypes.c_void_p(d_b), ctypes.c_void_p(d_c), ctypes.c_size_t(domain_size)))

    def close(self):
        if self.h:
            self.ctx.lib.zkpor_r1cs_destroy(self.h)
            self.h = None


NOT_PAUSED = 0xffffffff


class Solver:
    """the solver program of a compiled circuit on the device (include/zkpor.h zkpor_solver_*): r1cs.Solve of groth16.Prove
    (prover.go:269) as one launch per level over the matrices of `r1cs`"""

    def __init__(self, r1cs, container, ctx=None):
        """ctx: the context the solver's launches belong to (default: the R1CS's); another context of the same GPU = another worker"""
        self.ctx = ctx or r1cs.ctx; self.r1cs = r1cs
        buf = np.ascontiguousarray(np.frombuffer(container, dtype=np.uint8)) if not isinstance(container, np.ndarray) else np.ascontiguousarray(container, dtype=np.uint8)
        h = ctypes.c_void_p()
        self.ctx._ck(self.ctx.lib.zkpor_solver_create_on(self.ctx.h, r1cs.h, _p(buf), ctypes.c_size_t(buf.nbytes), ctypes.byref(h)))
        self.h = h

    def dims(self):
        d = (ctypes.c_uint64 * 7)()
        self.ctx._ck(self.ctx.lib.zkpor_solver_dims(self.h, d))
        return dict(zip(("instructions", "levels", "constraint_instructions", "hint_instructions", "skipped", "external_levels", "launches_last_run"), [int(x) for x in d]))

    def run(self, inputs, prefilled=None):
        """host-buffer form: (w, stats)"""
        inputs = _u64(inputs).reshape(-1, 4)
        ids = np.ascontiguousarray([i for i, _ in (prefilled or [])], dtype=np.uint32)
        vals = _u64(np.array([v for _, v in prefilled])).reshape(-1, 4) if prefilled else np.zeros((0, 4), np.uint64)
        w = np.zeros((self.r1cs.n_wires, 4), np.uint64)
        st = (ctypes.c_uint64 * 4)()
        self.ctx._ck(self.ctx.lib.zkpor_solver_run(self.h, _p(inputs), ctypes.c_size_t(inputs.shape[0]), _p(ids), _p(vals), ctypes.c_size_t(ids.shape[0]), _p(w), st))
        return w, dict(zip(("constraint_instructions", "hint_instructions", "skipped", "launches"), [int(x) for x in st]))

    def start_dev(self, d_w, n_inputs, d_known=None):
        paused = ctypes.c_uint32()
        self.ctx._ck(self.ctx.lib.zkpor_solver_start_dev(self.h, ctypes.c_void_p(d_w), ctypes.c_size_t(n_inputs), ctypes.c_void_p(d_known) if d_known else None, ctypes.byref(paused)))
        return paused.value

    def set_abc_dev(self, d_a, d_b, d_c):
        """the prove tail's a / b / c buffers: the Poseidon instructions of the next runs write their own rows (None, None, None = off)"""
        vp = ctypes.c_void_p
        self.ctx._ck(self.ctx.lib.zkpor_solver_set_abc_dev(self.h, vp(d_a) if d_a else None, vp(d_b) if d_b else None, vp(d_c) if d_c else None))

    def eval_abc_dev(self, d_w, d_a, d_b, d_c, domain_size):
        """a, b, c of every row the finished run has not written already"""
        vp = ctypes.c_void_p
        self.ctx._ck(self.ctx.lib.zkpor_solver_eval_abc_dev(self.h, vp(d_w), vp(d_a), vp(d_b), vp(d_c), ctypes.c_size_t(domain_size)))

    def prefetch_dev(self, d_w_next, n_inputs):
        """start the NEXT proof's ASYNC instructions (the CEX commitment chains) on the side stream; d_w_next must be what the next start_dev gets"""
        self.ctx._ck(self.ctx.lib.zkpor_solver_prefetch_dev(self.h, ctypes.c_void_p(d_w_next), ctypes.c_size_t(n_inputs)))

    def resume_dev(self):
        paused = ctypes.c_uint32()
        self.ctx._ck(self.ctx.lib.zkpor_solver_resume_dev(self.h, ctypes.byref(paused)))
        return paused.value

    def external_inputs(self, instr):
        n_in = ctypes.c_size_t(); n_out = ctypes.c_size_t()
        self.ctx._ck(self.ctx.lib.zkpor_solver_external_inputs(self.h, ctypes.c_uint32(instr), None, ctypes.c_size_t(0), ctypes.byref(n_in), ctypes.byref(n_out)))
        vals = np.zeros((n_in.value, 4), np.uint64)
        self.ctx._ck(self.ctx.lib.zkpor_solver_external_inputs(self.h, ctypes.c_uint32(instr), _p(vals), ctypes.c_size_t(n_in.value), ctypes.byref(n_in), ctypes.byref(n_out)))
        return vals, n_out.value

    def external_inputs_dev(self, instr, d_out, capacity):
        self.ctx._ck(self.ctx.lib.zkpor_solver_external_inputs_dev(self.h, ctypes.c_uint32(instr), ctypes.c_void_p(d_out), ctypes.c_size_t(capacity)))

    def external_outputs(self, instr, values):
        values = _u64(values).reshape(-1, 4)
        self.ctx._ck(self.ctx.lib.zkpor_solver_external_outputs(self.h, ctypes.c_uint32(instr), _p(values), ctypes.c_size_t(values.shape[0])))

    def close(self):
        if self.h:
            self.ctx.lib.zkpor_solver_destroy(self.h)
            self.h = None
