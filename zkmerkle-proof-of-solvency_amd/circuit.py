"""ctypes binding of the circuit compiler (host/circuit/*.hpp through libzkpor_host.so) and the end-to-end driver of groth16.Prove on the
device for a compiled circuit: BatchCreateUserCircuit.Define (circuit/batch_create_user_circuit.go:98-323) restated and compiled to the
constraint matrices + solver program the GPU backend executes — the stand-in, in an image without Go, for frontend.Compile
(src/keygen/main.go:30) + go/export_r1cs + go/export_solver.

    prove(...) below is src/prover/prover/prover.go:254-274 (assign -> NewWitness -> groth16.Prove) with every step after the assignment
    on the device: inputs up, solver program (zkpor_solver_*) until gnark's BSB22 placeholder, committed wires straight into
    zkpor_commit_dev, challenge hashed on the host (host/bsb22_challenge.hpp), resume, a / b / c = L.w, R.w, O.w (zkpor_r1cs_eval_dev),
    prove tail (zkpor_prove_tail_dev)."""
import ctypes
import json
import os

import numpy as np

import zkpor

_HOST = None


def host_lib():
    global _HOST
    if _HOST is None:
        L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libzkpor_host.so"))
        L.zkc_compile_batch_create_user.restype = ctypes.c_void_p
        L.zkc_synth_inputs.restype = ctypes.c_long
        L.zkc_census.restype = ctypes.c_char_p
        for name, t in (("zkc_coeff", ctypes.c_uint64), ("zkc_values", ctypes.c_uint64), ("zkc_level_ptr", ctypes.c_uint64), ("zkc_committed", ctypes.c_uint32),
                        ("zkc_in_l", ctypes.c_uint8), ("zkc_in_r", ctypes.c_uint8), ("zkc_solver_container", ctypes.c_uint8)):
            getattr(L, name).restype = ctypes.POINTER(t)
        for name, t in (("zkc_row_ptr", ctypes.c_uint64), ("zkc_cid", ctypes.c_uint32), ("zkc_wid", ctypes.c_uint32)):
            getattr(L, name).restype = ctypes.POINTER(t)
        _HOST = L
    return _HOST


def n_inputs(user_assets, all_assets, users):
    """elements of the assignment (public first, without the ONE wire): host/witness_assign.hpp"""
    return 1 + 5 + 114 * all_assets + users * (7 * user_assets + 5 * all_assets + 30)


def synth_inputs(user_assets, all_assets, users, seed=7, first_index=3):
    """a valid synthetic batch of that shape, assigned and in Montgomery form (host/circuit/synth_batch.hpp)"""
    L = host_lib()
    n = n_inputs(user_assets, all_assets, users)
    out = np.zeros((n, 4), np.uint64)
    err = ctypes.create_string_buffer(512)
    got = L.zkc_synth_inputs(user_assets, all_assets, users, ctypes.c_uint64(seed), first_index, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), err, 512)
    if got != n:
        raise RuntimeError("synth_inputs: " + err.value.decode())
    return out


class Circuit:
    """a compiled BatchCreateUserCircuit: matrices, solver program, commitment info (views into the library's memory)"""

    def __init__(self, user_assets, all_assets, users, inputs=None, commitment=None, poseidon_native=True):
        L = host_lib()
        err = ctypes.create_string_buffer(512)
        self.shape = (user_assets, all_assets, users)
        inp = np.ascontiguousarray(inputs, dtype=np.uint64) if inputs is not None else None
        cm = np.ascontiguousarray(commitment, dtype=np.uint64) if commitment is not None else None
        z = L.zkc_compile_batch_create_user(user_assets, all_assets, users, inp.ctypes.data_as(ctypes.c_void_p) if inp is not None else None,
                                            cm.ctypes.data_as(ctypes.c_void_p) if cm is not None else None, 1 if poseidon_native else 0, err, 512)
        if not z:
            raise RuntimeError("circuit: " + err.value.decode())
        self.z = ctypes.c_void_p(z)
        d = (ctypes.c_uint64 * 16)()
        L.zkc_dims(self.z, d)
        names = "n_wires n_public n_secret n_constraints n_coeff nnz_l nnz_r nnz_o n_instructions n_levels n_calldata n_committed commitment_wire no_l no_r container_bytes".split()
        self.dims = dict(zip(names, [int(x) for x in d]))
        for k, v in self.dims.items():
            setattr(self, k, v)
        self.census = json.loads(L.zkc_census(self.z).decode())
        self.interpreted = inputs is not None

    def _view(self, fn, n, shape=None, *args):
        p = getattr(host_lib(), fn)(self.z, *args)
        a = np.ctypeslib.as_array(p, shape=(n,))
        return a.reshape(shape) if shape else a

    def coeff(self):
        return self._view("zkc_coeff", 4 * self.n_coeff, (self.n_coeff, 4))

    def matrix(self, m):
        nnz = (self.nnz_l, self.nnz_r, self.nnz_o)[m]
        return (self._view("zkc_row_ptr", self.n_constraints + 1, None, m), self._view("zkc_cid", nnz, None, m), self._view("zkc_wid", nnz, None, m))

    def committed(self):
        return self._view("zkc_committed", self.n_committed)

    def infinity_masks(self):
        """(inf_a, inf_b) as gnark's pk.InfinityA / InfinityB: 1 where the wire appears in no L / R row"""
        in_l = self._view("zkc_in_l", self.n_wires); in_r = self._view("zkc_in_r", self.n_wires)
        return (in_l == 0).astype(np.uint8), (in_r == 0).astype(np.uint8)

    def values(self):
        """the interpreter's wire vector (only when compiled with inputs)"""
        assert self.interpreted
        return self._view("zkc_values", 4 * self.n_wires, (self.n_wires, 4))

    def level_sizes(self):
        lp = self._view("zkc_level_ptr", self.n_levels + 1)
        return np.diff(lp.astype(np.int64))

    def solver_container(self):
        return self._view("zkc_solver_container", self.container_bytes)

    def solve_host(self, inputs, commitment, threads=4, check_rows=True):
        """the compiled program on the host executor (host/solver_exec.hpp): the full wire vector"""
        w = np.zeros((self.n_wires, 4), np.uint64)
        err = ctypes.create_string_buffer(512)
        inp = np.ascontiguousarray(inputs, dtype=np.uint64); cm = np.ascontiguousarray(commitment, dtype=np.uint64)
        rc = host_lib().zkc_solve_host(self.z, inp.ctypes.data_as(ctypes.c_void_p), cm.ctypes.data_as(ctypes.c_void_p), threads, w.ctypes.data_as(ctypes.c_void_p),
                                       1 if check_rows else 0, err, 512)
        if rc != 0:
            raise RuntimeError("host executor: %d %s" % (rc, err.value.decode()))
        return w

    def solve_host_with(self, container, inputs, commitment, threads=4, check_rows=True):
        """the host executor over ANOTHER program for this circuit's matrices (tests rearrange the levels)"""
        buf = np.ascontiguousarray(np.frombuffer(bytes(container), dtype=np.uint8))
        inp = np.ascontiguousarray(inputs, dtype=np.uint64); cm = np.ascontiguousarray(commitment, dtype=np.uint64)
        w = np.zeros((self.n_wires, 4), np.uint64)
        err = ctypes.create_string_buffer(512)
        rc = host_lib().zkc_solve_host_with(self.z, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes), inp.ctypes.data_as(ctypes.c_void_p),
                                            cm.ctypes.data_as(ctypes.c_void_p), threads, w.ctypes.data_as(ctypes.c_void_p), 1 if check_rows else 0, err, ctypes.c_size_t(512))
        if rc != 0:
            raise RuntimeError(f"host executor: {err.value.decode()}")
        return w

    def close(self):
        if self.z:
            host_lib().zkc_free(self.z)
            self.z = None


# ---- one compile per NODE: the compiled circuit handed to the other ranks through /dev/shm ----
# `bench.py --gpus 8` is eight processes; the compile (10 s, ~13 GB of matrices) is the same for every rank of a tier.  Rank 0 compiles and
# writes the arrays every consumer reads — coefficient table, three CSR matrices, solver container, commitment info, level sizes — as .npy files
# under /dev/shm; the others map them (no copy: the page cache holds ONE set) behind the same read interface as `Circuit`.
_SHARED_ARRAYS = ("coeff", "row_ptr0", "cid0", "wid0", "row_ptr1", "cid1", "wid1", "row_ptr2", "cid2", "wid2", "committed", "in_l", "in_r", "level_ptr", "container")


def shared_dir(tag):
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    return os.path.join(base, "zkpor_circuit_" + "".join(ch if ch.isalnum() or ch in "-_" else "_" for ch in str(tag)))


def export_shared(circuit, tag):
    """write a compiled Circuit's arrays for SharedCircuit(tag); returns the directory"""
    d = shared_dir(tag)
    os.makedirs(d, exist_ok=True)
    arrays = {"coeff": circuit.coeff(), "committed": circuit.committed(), "container": circuit.solver_container(),
              "in_l": circuit._view("zkc_in_l", circuit.n_wires), "in_r": circuit._view("zkc_in_r", circuit.n_wires),
              "level_ptr": circuit._view("zkc_level_ptr", circuit.n_levels + 1)}
    for m in range(3):
        rp, cid, wid = circuit.matrix(m)
        arrays[f"row_ptr{m}"] = rp; arrays[f"cid{m}"] = cid; arrays[f"wid{m}"] = wid
    for name in _SHARED_ARRAYS:
        np.save(os.path.join(d, name + ".npy"), np.ascontiguousarray(arrays[name]))
    with open(os.path.join(d, "meta.json.tmp"), "w") as f:
        json.dump({"shape": list(circuit.shape), "dims": circuit.dims, "census": circuit.census}, f)
    os.replace(os.path.join(d, "meta.json.tmp"), os.path.join(d, "meta.json"))       # the meta file appears last: a reader that finds it finds everything
    return d


def unlink_shared(tag):
    d = shared_dir(tag)
    if os.path.isdir(d):
        for n in os.listdir(d):
            try:
                os.unlink(os.path.join(d, n))
            except OSError:
                pass
        try:
            os.rmdir(d)
        except OSError:
            pass


class SharedCircuit:
    """the read side: a circuit another process compiled (export_shared), mapped read-only.  Everything DeviceCircuit, the key synthesis and the
    checks read from a Circuit; no interpreter, no host executor (those stay with the process that compiled)"""

    def __init__(self, tag):
        d = shared_dir(tag)
        with open(os.path.join(d, "meta.json")) as f:
            meta = json.load(f)
        self.shape = tuple(meta["shape"]); self.dims = meta["dims"]; self.census = meta["census"]
        for k, v in self.dims.items():
            setattr(self, k, v)
        self._a = {name: np.load(os.path.join(d, name + ".npy"), mmap_mode="r") for name in _SHARED_ARRAYS}
        self.interpreted = False
        self.z = None

    def coeff(self):
        return self._a["coeff"]

    def matrix(self, m):
        return (self._a[f"row_ptr{m}"], self._a[f"cid{m}"], self._a[f"wid{m}"])

    def committed(self):
        return self._a["committed"]

    def infinity_masks(self):
        return (np.asarray(self._a["in_l"]) == 0).astype(np.uint8), (np.asarray(self._a["in_r"]) == 0).astype(np.uint8)

    def level_sizes(self):
        return np.diff(np.asarray(self._a["level_ptr"]).astype(np.int64))

    def solver_container(self):
        return self._a["container"]

    def close(self):
        self._a = {}


def default_commitment():
    """the value the interpreter gives the commitment wire when none is passed (frontend.hpp commitment_value_), Montgomery form"""
    out = np.zeros(4, np.uint64)
    host_lib().zkh_fr_from_canon(np.array([0x5eedc0de, 0, 0, 0], np.uint64).ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(1))
    return out


class DeviceCircuit:
    """the compiled circuit resident on the device: matrices (zkpor_r1cs_*) + solver program (zkpor_solver_*)"""

    def __init__(self, ctx, circuit, share=None):
        """share: another DeviceCircuit of the same circuit on the same GPU — its matrices are used (read-only), only the program is loaded
        again, bound to `ctx` (one solver per worker context: zkpor_solver_create_on)"""
        self.ctx = ctx; self.circuit = circuit
        self.owns_r1cs = share is None
        if share is None:
            self.r1cs = zkpor.R1CS(ctx, circuit.n_constraints, circuit.n_wires, circuit.coeff())
            for m in range(3):
                self.r1cs.set_matrix(m, *circuit.matrix(m))
        else:
            self.r1cs = share.r1cs
        self.solver = zkpor.Solver(self.r1cs, circuit.solver_container(), ctx=ctx)

    def close(self):
        self.solver.close()
        if self.owns_r1cs:
            self.r1cs.close()


def bsb22_challenge(ctx, commitment_affine):
    """the BSB22 hint's output for one commitment and no public committed wires, Montgomery limbs (host/bsb22_challenge.hpp)"""
    be = np.zeros(64, np.uint8)
    ctx._ck(ctx.lib.zkpor_g1_marshal(zkpor._p(np.ascontiguousarray(commitment_affine)), zkpor._p(be)))
    out = (ctypes.c_uint8 * 32)()
    if host_lib().zkh_bsb22_challenge(bytes(be), None, ctypes.c_size_t(0), out) != 0:
        raise RuntimeError("bsb22 challenge")
    canon = np.frombuffer(bytes(out)[::-1], dtype=np.uint64).copy()
    mont = np.zeros(4, np.uint64)
    host_lib().zkh_fr_from_canon(canon.ctypes.data_as(ctypes.c_void_p), mont.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(1))
    return mont


def stage_inputs(ctx, dc, d_w, d_inputs_or_host):
    """wire 0 = ONE and the assignment into d_w (what solve_on_device does first; call it yourself to prefetch: Solver.prefetch_dev)"""
    c = dc.circuit
    n_in = c.n_public + c.n_secret
    one = np.array([0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f], np.uint64)
    vp = ctypes.c_void_p
    ctx._ck(ctx.lib.zkpor_dev_upload(ctx.h, vp(d_w), zkpor._p(one), ctypes.c_size_t(32)))
    if isinstance(d_inputs_or_host, int):
        ctx._ck(ctx.lib.zkpor_dev_copy(ctx.h, vp(d_w + 32), vp(d_inputs_or_host), ctypes.c_size_t(32 * (n_in - 1))))
    else:
        a = np.ascontiguousarray(d_inputs_or_host, dtype=np.uint64)
        ctx._ck(ctx.lib.zkpor_dev_upload(ctx.h, vp(d_w + 32), zkpor._p(a), ctypes.c_size_t(a.nbytes)))
    return n_in


def solve_on_device(ctx, dc, pk, d_w, d_cv, d_inputs_or_host, timings=None, staged=False):
    """inputs -> the full wire vector in d_w, serving the BSB22 commitment: returns (commitment, pok, challenge).  d_inputs_or_host: a device
    pointer (int) to n_inputs Montgomery elements, or a host array (uploaded here).  d_w: n_wires x 32 B, d_cv: (1 + n_committed) x 32 B — the
    placeholder's inputs: the commitment index, then the committed wires (the values zkpor_commit_dev sums start at d_cv + 32)."""
    import time
    c = dc.circuit
    n_in = c.n_public + c.n_secret
    one = np.array([0xac96341c4ffffffb, 0x36fc76959f60cd29, 0x666ea36f7879462e, 0x0e0a77c19a07df2f], np.uint64)
    lib = ctx.lib
    vp = ctypes.c_void_p
    t0 = time.perf_counter()
    if not staged:
        stage_inputs(ctx, dc, d_w, d_inputs_or_host)
    s = dc.solver
    paused = s.start_dev(d_w, n_in)
    t1 = time.perf_counter()
    com = pok = ch = None
    while paused != zkpor.NOT_PAUSED:
        s.external_inputs_dev(paused, d_cv, c.n_committed + 1)
        com = np.empty(8, np.uint64); pok = np.empty(8, np.uint64)
        ctx._ck(lib.zkpor_commit_dev(ctx.h, pk.h, vp(d_cv + 32), ctypes.c_size_t(c.n_committed), zkpor._p(com), zkpor._p(pok)))
        ch = bsb22_challenge(ctx, com)
        t2 = time.perf_counter()
        s.external_outputs(paused, ch.reshape(1, 4))
        paused = s.resume_dev()
    t3 = time.perf_counter()
    if timings is not None:
        timings.update({"solve_phase1_ms": (t1 - t0) * 1e3, "commit_ms": ((t2 - t1) * 1e3) if com is not None else 0.0,
                        "solve_phase2_ms": ((t3 - t2) * 1e3) if com is not None else 0.0})
    return com, pok, ch
