"""ctypes binding of libzkpor.so (include/zkpor.h) — the Python-side stand-in for the cgo shim a maintainer adds to
the reference's src/prover.  There is NO fallback: if the HIP library is missing or no gfx950 device
is usable, construction raises."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkpor.so")

G1_A, G1_B, G1_K, G1_Z, G1_COMMIT_BASIS, G1_COMMIT_BASIS_SIGMA = range(6)
G2_B = 0
Z_ORDER_BITREV, Z_ORDER_NATURAL = 0, 1


class ZkporError(RuntimeError):
    pass


_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZkporError(f"{LIB_PATH} is not built (run __graft_entry__.build()); there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
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

    def phase_ms(self, name):
        calls = ctypes.c_uint64()
        ms = self.lib.zkpor_phase_ms(self.h, name.encode(), ctypes.byref(calls))
        return ms, calls.value

    def phase_reset(self):
        self.lib.zkpor_phase_reset(self.h)

    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    def fill_fr(self, buf, n, seed, kind=0):
        self._ck(self.lib.zkpor_dev_fill_fr(self.h, ctypes.c_void_p(buf.ptr), ctypes.c_size_t(n), ctypes.c_uint64(seed), ctypes.c_int(kind)))

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
