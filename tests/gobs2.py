"""Test-side, INDEPENDENT restatement (pure Python) of the two third-party formats under the witness row
(src/witness/witness/witness.go:215-232: base64(s2(gob(BatchCreateUserWitness)))): Go's encoding/gob wire format as its
package documentation specifies it, and the s2 / Snappy block format.  The product's codec is C++ (host/witness_codec.hpp);
this file exists so that neither side is checked against itself.  Also the deterministic synthetic witness both sides build."""
import struct

M64 = (1 << 64) - 1


# ---------------------------------------------------------------------------------------------- gob
def g_uint(v):
    if v < 128:
        return bytes([v])
    b = v.to_bytes((v.bit_length() + 7) // 8, "big")
    return bytes([256 - len(b)]) + b


def g_int(i):
    return g_uint((~(i << 1)) & M64 if i < 0 else i << 1)


def g_bytes(b):
    return g_uint(len(b)) + bytes(b)


class R:
    def __init__(self, b):
        self.b, self.o = b, 0

    def eof(self):
        return self.o >= len(self.b)

    def uint(self):
        x = self.b[self.o]; self.o += 1
        if x < 128:
            return x
        n = 256 - x
        assert 1 <= n <= 8
        v = int.from_bytes(self.b[self.o:self.o + n], "big"); self.o += n
        return v

    def sint(self):
        u = self.uint()
        return ~(u >> 1) if u & 1 else u >> 1

    def bytes(self):
        n = self.uint()
        assert self.o + n <= len(self.b)
        v = bytes(self.b[self.o:self.o + n]); self.o += n
        return v


BUILTIN = {1: "uint", 2: "int", 3: "uint", 4: "uint", 5: "bytes", 6: "bytes"}


def _common(m):
    name, tid, f = "", 0, -1
    while True:
        d = m.uint()
        if not d:
            return name, tid
        f += d
        if f == 0:
            name = m.bytes().decode()
        elif f == 1:
            tid = m.sint()
        else:
            raise ValueError("CommonType field")


def _define(m):
    wf = -1
    d = m.uint(); wf += d
    t = {"kind": ["array", "slice", "struct", "map", "gobenc", "gobenc", "gobenc"][wf], "fields": []}
    f = -1
    while True:
        d = m.uint()
        if not d:
            break
        f += d
        if f == 0:
            t["name"], t["id"] = _common(m)
        elif t["kind"] in ("array", "slice") and f == 1:
            t["elem"] = m.sint()
        elif t["kind"] == "array" and f == 2:
            t["len"] = m.sint()
        elif t["kind"] == "struct" and f == 1:
            for _ in range(m.uint()):
                ff, nm, fid = -1, "", 0
                while True:
                    d3 = m.uint()
                    if not d3:
                        break
                    ff += d3
                    if ff == 0:
                        nm = m.bytes().decode()
                    else:
                        fid = m.sint()
                t["fields"].append((nm, fid))
        else:
            raise ValueError("type description field")
    assert m.uint() == 0 and m.eof()
    t.setdefault("len", 0)
    return t


def _value(types, tid, m):
    if tid in BUILTIN:
        return {"uint": m.uint, "int": m.sint, "bytes": m.bytes}[BUILTIN[tid]]()
    t = types[tid]
    if t["kind"] == "gobenc":
        return m.bytes()
    if t["kind"] in ("array", "slice"):
        n = m.uint()
        if t["kind"] == "array":
            assert n == t["len"]
        return [_value(types, t["elem"], m) for _ in range(n)]
    out, f = {}, -1
    while True:
        d = m.uint()
        if not d:
            return out
        f += d
        name, fid = t["fields"][f]
        out[name] = _value(types, fid, m)


def gob_decode(stream):
    """-> (value tree, types); structs are dicts holding only the fields that were sent"""
    r = R(stream)
    types = {}
    while not r.eof():
        n = r.uint()
        m = R(stream[r.o:r.o + n]); r.o += n
        tid = m.sint()
        if tid < 0:
            assert -tid >= 64 and -tid not in types
            types[-tid] = _define(m)
            continue
        if tid in types and types[tid]["kind"] == "struct":
            v = _value(types, tid, m)
        else:
            assert m.uint() == 0
            v = _value(types, tid, m)
        assert m.eof() and r.eof()
        return v, types
    raise ValueError("no value")


def _typedef(tid, t):
    common = (g_uint(1) + g_bytes(t["name"].encode()) + g_uint(1) if t.get("name") else g_uint(2)) + g_int(tid) + g_uint(0)
    k = t["kind"]
    if k == "struct":
        body = g_uint(3) + g_uint(1) + common
        if t["fields"]:
            body += g_uint(1) + g_uint(len(t["fields"])) + b"".join(g_uint(1) + g_bytes(n.encode()) + g_uint(1) + g_int(i) + g_uint(0) for n, i in t["fields"])
        body += g_uint(0)
    elif k == "slice":
        body = g_uint(2) + g_uint(1) + common + g_uint(1) + g_int(t["elem"]) + g_uint(0)
    elif k == "array":
        body = g_uint(1) + g_uint(1) + common + g_uint(1) + g_int(t["elem"]) + g_uint(1) + g_int(t["len"]) + g_uint(0)
    else:
        body = g_uint(5) + g_uint(1) + common + g_uint(0)
    return g_int(-tid) + body + g_uint(0)


def _enc(types, tid, v, send_zero):
    if tid in BUILTIN:
        return {"uint": g_uint, "int": g_int, "bytes": g_bytes}[BUILTIN[tid]](v)
    t = types[tid]
    if t["kind"] == "gobenc":
        return g_bytes(v)
    if t["kind"] in ("array", "slice"):
        return g_uint(len(v)) + b"".join(_enc(types, t["elem"], x, send_zero) for x in v)
    out, last = b"", -1
    for i, (name, fid) in enumerate(t["fields"]):
        if name not in v:
            continue
        x = v[name]
        if not send_zero and (x == 0 or x == b"" or x == [] or x == {}):
            continue
        out += g_uint(i - last) + _enc(types, fid, x, send_zero); last = i
    return out + g_uint(0)


def gob_encode(types, order, top, value, send_zero=False):
    """types: {id: description}; order: ids in the order their definitions are sent"""
    msg = lambda p: g_uint(len(p)) + p
    return b"".join(msg(_typedef(i, types[i])) for i in order) + msg(g_int(top) + _enc(types, top, value, send_zero))


# ---------------------------------------------------------------------------------------------- s2 / snappy block
def s2_decode(b):
    i, want, shift = 0, 0, 0
    while True:
        x = b[i]; i += 1
        want |= (x & 0x7f) << shift
        if not x & 0x80:
            break
        shift += 7
    out = bytearray()
    last = 1
    while i < len(b):
        tag = b[i]
        k = tag & 3
        if k == 0:
            l = tag >> 2; i += 1
            if l >= 60:
                e = l - 59
                l = int.from_bytes(b[i:i + e], "little"); i += e
            l += 1
            out += b[i:i + l]; i += l
            continue
        if k == 1:
            off = ((tag & 0xe0) << 3) | b[i + 1]
            ln = (tag >> 2) & 7
            i += 2
            if off == 0:
                off = last
                if ln == 5:
                    ln = b[i] + 4; i += 1
                elif ln == 6:
                    ln = int.from_bytes(b[i:i + 2], "little") + 256; i += 2
                elif ln == 7:
                    ln = int.from_bytes(b[i:i + 3], "little") + 65536; i += 3
            ln += 4
        elif k == 2:
            ln = (tag >> 2) + 1; off = int.from_bytes(b[i + 1:i + 3], "little"); i += 3
        else:
            ln = (tag >> 2) + 1; off = int.from_bytes(b[i + 1:i + 5], "little"); i += 5
        assert 0 < off <= len(out)
        last = off
        for _ in range(ln):
            out.append(out[-off])
    assert len(out) == want
    return bytes(out)


def s2_literal_block(src):
    out = bytearray()
    v = len(src)
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80); v >>= 7
    out.append(v)
    i = 0
    while i < len(src):
        k = min(len(src) - i, 65536)
        l = k - 1
        if l < 60:
            out.append(l << 2)
        elif l < 256:
            out += bytes([60 << 2, l])
        else:
            out += bytes([61 << 2]) + struct.pack("<H", l)
        out += src[i:i + k]; i += k
    return bytes(out)


# ---------------------------------------------------------------------------------------------- the synthetic witness
class _Mix:
    def __init__(self, seed):
        self.s = seed & M64

    def __call__(self):
        self.s = (self.s + 0x9e3779b97f4a7c15) & M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M64
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M64
        return z ^ (z >> 31)

    def bytes(self, n):
        return bytes(self() & 0xff for _ in range(n))


def _big(v):  # math/big GobEncode: version 1, positive
    return b"\x02" + (v.to_bytes((v.bit_length() + 7) // 8, "big") if v else b"")


def synth_witness(seed, users, assets_per_user, cex_assets):
    """mirror of host_capi.cpp synth_witness: the FULL value (zero fields included), big ints as GobEncode bytes or None"""
    m = _Mix(seed)
    w = {"BatchCommitment": m.bytes(32), "AccountTreeRoot": m.bytes(32), "BeforeCEXAssetsCommitment": m.bytes(32),
         "AfterCEXAssetsCommitment": m.bytes(32)}
    w["MinAccountIndex"] = m() & 0xffffffff; w["MaxAccountIndex"] = m() & 0xffffffff
    w["BeforeCexAssets"] = []
    for i in range(cex_assets):
        c = {"TotalEquity": m(), "TotalDebt": m() >> 20, "BasePrice": m() >> 40, "Symbol": b"" if i % 3 == 0 else ("sym%d" % i).encode(), "Index": i}
        c["LoanCollateral"] = m() >> 8; c["MarginCollateral"] = m() if i % 2 else 0; c["PortfolioMarginCollateral"] = m() >> 1
        for l, name in enumerate(("LoanRatios", "MarginRatios", "PortfolioMarginRatios")):
            arr = []
            for t in range(12):
                if l == 2 and i % 4 == 0:
                    arr.append({"BoundaryValue": None, "Ratio": 0, "PrecomputedValue": None}); continue
                hi = m() >> 10; lo = m()
                tr = {"BoundaryValue": _big((hi << 64) | lo), "Ratio": m() % 101}
                tr["PrecomputedValue"] = _big(0 if t == 0 else m() * t)
                arr.append(tr)
            c[name] = arr
        w["BeforeCexAssets"].append(c)
    w["CreateUserOps"] = []
    for u in range(users):
        assets = []
        for a in range(assets_per_user):
            x = {"Index": (a * 7 + u) % 500, "Equity": m() >> 24}
            x["Debt"] = m() >> 30 if a % 2 else 0
            x["Loan"] = m() >> 50; x["Margin"] = 0; x["PortfolioMargin"] = m()
            assets.append(x)
        op = {"Assets": assets, "AccountIndex": 1000 + u, "AccountIdHash": m.bytes(32)}
        op["AccountProof"] = [m.bytes(32) for _ in range(28)]
        w["CreateUserOps"].append(op)
    return w


ZERO = {"BatchCommitment": b"", "AccountTreeRoot": b"", "BeforeCEXAssetsCommitment": b"", "AfterCEXAssetsCommitment": b"",
        "MinAccountIndex": 0, "MaxAccountIndex": 0, "BeforeCexAssets": [], "CreateUserOps": []}


def normalise(v, full):
    """fill a decoded tree (fields that were omitted on the wire) up to the shape of `full` so the two compare with =="""
    if isinstance(full, dict):
        out = {}
        for k, fv in full.items():
            if v is not None and k in v:
                out[k] = normalise(v[k], fv)
            else:
                out[k] = _zero_like(fv)
        return out
    if isinstance(full, list):
        if v is None:
            return _zero_like(full)
        return [normalise(x, f) for x, f in zip(v, full)] if len(v) == len(full) else v
    return v


def _zero_like(fv):
    if isinstance(fv, dict):
        return {k: _zero_like(x) for k, x in fv.items()}
    if isinstance(fv, list):
        # an omitted array of structs decodes as all-zero elements; an omitted slice as empty — the caller compares against `full`,
        # whose all-zero arrays are written out explicitly
        return [_zero_like(x) for x in fv] if fv and isinstance(fv[0], (dict, bytes)) and len(fv) in (12, 28) else []
    if isinstance(fv, (bytes, bytearray)):
        return b""
    if fv is None:
        return None
    return 0
