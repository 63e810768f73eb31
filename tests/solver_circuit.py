"""Test-side builder of a small R1CS WITH its solver program, in the shapes gnark compiles BatchCreateUserCircuit into (SURVEY.md §8 a8 /
Appendix B): multiplication wires, assertions, inverses (division path of the solver), ToBinary (NBits hint + booleanity + recomposition),
IsZero (InvZero hint), the circuit's own IntegerDivision hint (circuit/utils.go:103-110) with its q * b = a - rem constraint, 16-bit range-check
decompositions (DecomposeHint), and x^5 S-box chains with linear mixing between them (the Poseidon gadget's shape).  The builder carries the
VALUE of every wire in Python integers, so the expected wire vector exists independently of the executor; it emits the flat r1cs container
(tests/r1cs_container.py) and the solver container (layout: host/solver_exec.hpp header), with the levels computed from the wire dependencies."""
import struct

import numpy as np

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
MONT = (1 << 256) % R


def to_mont_limbs(vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = (v % R) * MONT % R
        for k in range(4):
            out[i, k] = (m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def from_mont_limbs(arr):
    """python integers of Montgomery limb rows"""
    inv = pow(MONT, R - 2, R)
    return [sum(int(x) << (64 * k) for k, x in enumerate(row)) * inv % R for row in np.asarray(arr, dtype=np.uint64).reshape(-1, 4)]


class Builder:
    def __init__(self, public, secret):
        """wire 0 = ONE, then the public inputs, then the secret ones (gnark's order)"""
        self.val = [1] + [v % R for v in public] + [v % R for v in secret]
        self.n_public = 1 + len(public)
        self.n_secret = len(secret)
        self.level_of_wire = [0] * len(self.val)
        self.coeffs = {}          # value -> id
        self.rows = []            # (L, R, O) as lists of (coeff id, wire)
        self.instr = []           # (kind, arg, level, tag)
        self.calldata = []
        self.hint_names = []
        self.cid(0); self.cid(1)

    # ---- linear expressions: dict wire -> coefficient
    def cid(self, c):
        c %= R
        if c not in self.coeffs:
            self.coeffs[c] = len(self.coeffs)
        return self.coeffs[c]

    def wire(self, i, c=1):
        return {i: c % R}

    def const(self, c):
        return {0: c % R}

    def add(self, *es):
        out = {}
        for e in es:
            for w, c in e.items():
                out[w] = (out.get(w, 0) + c) % R
        return {w: c for w, c in out.items() if c}

    def scale(self, e, k):
        return {w: c * k % R for w, c in e.items() if c * k % R}

    def sub(self, a, b):
        return self.add(a, self.scale(b, R - 1))

    def eval(self, e):
        return sum(c * self.val[w] for w, c in e.items()) % R

    def _terms(self, e):
        return [(self.cid(c), w) for w, c in sorted(e.items())]

    def _lvl(self, *es):
        return max([self.level_of_wire[w] for e in es for w in e] + [0])

    def new_wire(self, value, level):
        self.val.append(value % R)
        self.level_of_wire.append(level)
        return len(self.val) - 1

    # ---- constraints
    def mul(self, a, b, tag="mul"):
        """new wire x with a * b = x"""
        lvl = self._lvl(a, b) + 1
        x = self.new_wire(self.eval(a) * self.eval(b), lvl)
        self.rows.append((self._terms(a), self._terms(b), self._terms(self.wire(x))))
        self.instr.append((0, len(self.rows) - 1, lvl, tag))
        return x

    def assert_mul(self, a, b, c, tag="assert"):
        lvl = self._lvl(a, b, c) + 1
        self.rows.append((self._terms(a), self._terms(b), self._terms(c)))
        self.instr.append((0, len(self.rows) - 1, lvl, tag))

    def inverse(self, a):
        """new wire x with a * x = 1 (the unknown sits on the R side: the solver divides)"""
        lvl = self._lvl(a) + 1
        x = self.new_wire(pow(self.eval(a), R - 2, R), lvl)
        self.rows.append((self._terms(a), self._terms(self.wire(x)), self._terms(self.const(1))))
        self.instr.append((0, len(self.rows) - 1, lvl, "inverse"))
        return x

    def div_left(self, num, den):
        """new wire x with x * den = num (unknown on the L side, scaled by 3 to exercise the coefficient inverse)"""
        lvl = self._lvl(num, den) + 1
        v = self.eval(num) * pow(self.eval(den), R - 2, R) % R
        x = self.new_wire(v * pow(3, R - 2, R), lvl)
        self.rows.append((self._terms(self.wire(x, 3)), self._terms(den), self._terms(num)))
        self.instr.append((0, len(self.rows) - 1, lvl, "div"))
        return x

    def hint(self, name, inputs, out_values, tag=None):
        if name not in self.hint_names:
            self.hint_names.append(name)
        lvl = self._lvl(*inputs) + 1
        outs = [self.new_wire(v, lvl) for v in out_values]
        off = len(self.calldata)
        self.calldata += [self.hint_names.index(name), len(inputs), len(outs)] + outs
        for e in inputs:
            t = self._terms(e)
            self.calldata.append(len(t))
            for cid, w in t:
                self.calldata += [cid, w]
        self.instr.append((1, off, lvl, tag or name))
        return outs

    # ---- gadgets
    def to_binary(self, e, n):
        v = self.eval(e)
        bits = self.hint("NBits", [e], [(v >> i) & 1 for i in range(n)])
        for b in bits:
            self.assert_mul(self.wire(b), self.sub(self.const(1), self.wire(b)), self.const(0), "bool")
        self.assert_mul(self.const(1), self.add(*[self.wire(b, 1 << i) for i, b in enumerate(bits)]), e, "recompose")
        return bits

    def is_zero(self, e):
        v = self.eval(e)
        (inv,) = self.hint("InvZero", [e], [pow(v, R - 2, R) if v else 0])
        m = self.mul(self.scale(e, R - 1), self.wire(inv), "iszero_m")      # m' = -e * inv ; m = 1 + m'
        self.assert_mul(e, self.add(self.const(1), self.wire(m)), self.const(0), "iszero_chk")
        return self.add(self.const(1), self.wire(m))

    def integer_division(self, a, b):
        va, vb = self.eval(a), self.eval(b)
        q, rem = self.hint("IntegerDivision", [a, b], [va // vb, va % vb])
        self.assert_mul(self.wire(q), b, self.sub(a, self.wire(rem)), "divmod")
        return q, rem

    def range_check(self, e, bits, limb=16):
        v = self.eval(e)
        n = (bits + limb - 1) // limb
        limbs = self.hint("DecomposeHint", [self.const(bits), self.const(limb), e], [(v >> (limb * i)) & ((1 << limb) - 1) for i in range(n)])
        self.assert_mul(self.const(1), self.add(*[self.wire(l, 1 << (limb * i)) for i, l in enumerate(limbs)]), e, "limbs")
        return limbs

    def sbox(self, e):
        x2 = self.mul(e, e, "sbox")
        x4 = self.mul(self.wire(x2), self.wire(x2), "sbox")
        return self.mul(self.wire(x4), e, "sbox")

    # ---- containers
    def levels(self):
        by = {}
        for i, (_, _, lvl, _) in enumerate(self.instr):
            by.setdefault(lvl, []).append(i)
        return [by[k] for k in sorted(by)]

    def tables(self):
        """(coefficient table as Montgomery limbs, [(row_ptr, coeff ids, wire ids) for L, R, O]) — the arguments of zkpor_r1cs_create / set_matrix"""
        table = [0] * len(self.coeffs)
        for v, i in self.coeffs.items():
            table[i] = v
        mats = []
        for side in range(3):
            ptr, cids, wids = [0], [], []
            for row in self.rows:
                for cid, w in row[side]:
                    cids.append(cid); wids.append(w)
                ptr.append(len(cids))
            mats.append((np.array(ptr, dtype=np.uint64), np.array(cids, dtype=np.uint32), np.array(wids, dtype=np.uint32)))
        return to_mont_limbs(table), mats

    def r1cs_bytes(self):
        import r1cs_container
        table, mats = self.tables()
        return r1cs_container.write(len(self.rows), len(self.val), self.n_public, self.n_secret, table, mats)

    def solver_bytes(self, levels=None, skip_tags=()):
        levels = self.levels() if levels is None else levels
        out = bytearray(b"ZKPSOLV\x01")
        out += struct.pack("<4Q", len(self.instr), len(levels), len(self.hint_names), len(self.calldata))
        for n in self.hint_names:
            out += struct.pack("<I", len(n)) + n.encode()
        out += b"\0" * (-len(out) % 8)
        kinds = [(2 if tag in skip_tags else k) for k, _, _, tag in self.instr]
        out += np.array(kinds, dtype="<u4").tobytes() + np.array([a for _, a, _, _ in self.instr], dtype="<u4").tobytes()
        out += b"\0" * (-len(out) % 8)
        ptr = [0]
        flat = []
        for lv in levels:
            flat += lv
            ptr.append(len(flat))
        out += np.array(ptr, dtype="<u8").tobytes() + np.array(flat, dtype="<u4").tobytes()
        out += b"\0" * (-len(out) % 8)
        out += np.array(self.calldata, dtype="<u4").tobytes()
        return bytes(out)

    def wires_of_tag(self, tag):
        """output wires of the R1C instructions carrying `tag` (each solves exactly the O-side wire)"""
        out = []
        for k, a, _, t in self.instr:
            if t == tag and k == 0:
                out.append(self.rows[a][2][0][1])
        return out


def demo_circuit(seed=1, n_users=6, chain=True):
    """a miniature of the real circuit's structure: per "user" a 64-bit balance range-checked and bit-decomposed, an integer division by a price,
    a zero test, two rounds of a width-3 S-box permutation with linear mixing, an inverse, a division.  chain=True threads one accumulator
    through the users (deep levels, as the running CEX totals do); chain=False leaves the users independent (wide levels, as the per-user
    Merkle / commitment gadgets are)"""
    rng = np.random.default_rng(seed)
    balances = [int(rng.integers(1, 1 << 62)) for _ in range(n_users)]
    prices = [int(rng.integers(1, 1 << 20)) for _ in range(n_users)]
    ids = [int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) for _ in range(n_users)]
    b = Builder(public=[123456789], secret=balances + prices + ids + [0])
    base = b.n_public
    zero_in = base + 3 * n_users
    acc = b.wire(1)
    for u in range(n_users):
        bal, price, uid = b.wire(base + u), b.wire(base + n_users + u), b.wire(base + 2 * n_users + u)
        b.range_check(bal, 64)
        bits = b.to_binary(price, 20)
        q, rem = b.integer_division(bal, price)
        b.range_check(b.wire(rem), 32)
        z = b.is_zero(b.sub(bal, b.wire(q))) if u % 2 else b.is_zero(b.wire(zero_in))
        st = [uid, b.add(acc, b.wire(q)), b.add(b.wire(bits[0]), z, b.const(7))]
        for rnd in range(2):
            st = [b.wire(b.sbox(b.add(s, b.const(1000 * rnd + i)))) for i, s in enumerate(st)]
            st = [b.add(b.scale(st[0], 2 + i), b.scale(st[1], 3 + i), b.scale(st[2], 5 + i)) for i in range(3)]
        inv = b.inverse(b.add(st[0], b.const(1)))
        d = b.div_left(st[1], b.add(b.wire(inv), b.const(2)))
        nxt = b.add(b.wire(d), st[2])
        if chain:
            acc = nxt
        else:
            b.assert_mul(nxt, b.const(1), nxt, "tail")
    b.assert_mul(acc, b.const(1), acc, "tail")
    return b


def poseidon_gadget(b, state, rp, rc, mds, record):
    """the in-circuit Poseidon permutation over len(state) lanes (plain HADES as oracle/poseidon.hpp poseidon_permute: round constants, S-box
    x^5 on every lane in the 4 + 4 full rounds and on lane 0 in the rp partial ones, MDS): three multiplication wires per S-box (x^2, x^4,
    x^5, tagged "sbox"), everything else linear.  `record` receives (input state values, [wire ids of the S-boxes in round order, 3 each]) —
    the slot order of zkpor_witgen_poseidon_trace_dev.  Returns the output state as linear expressions."""
    t = len(state)
    inputs = [b.eval(e) for e in state]
    wires = []
    k = 0
    for r in range(8 + rp):
        state = [b.add(state[i], b.const(rc[k + i])) for i in range(t)]
        k += t
        for i in (range(t) if (r < 4 or r >= 4 + rp) else [0]):
            x = state[i]
            x2 = b.mul(x, x, "sbox"); x4 = b.mul(b.wire(x2), b.wire(x2), "sbox"); x5 = b.mul(b.wire(x4), x, "sbox")
            wires += [x2, x4, x5]
            state[i] = b.wire(x5)
        state = [b.add(*[b.scale(state[j], mds[i * t + j]) for j in range(t)]) for i in range(t)]
    record.append((inputs, wires))
    return state


def poseidon_circuit(params, seed=1, paths=3, depth=4, wide=2):
    """`paths` Merkle paths of `depth` width-3 hashes each (the leaf and the siblings are secret inputs, the root is asserted against a public
    input: a chain of permutations — deep) and `wide` independent width-13 permutations of secret inputs (one block of the account sponge),
    their lane-1 outputs asserted against public inputs.  params = {t: (rp, rc ints, mds ints)} (oracle.poseidon_params).  Returns
    (builder, {t: [(input state values, sbox wire ids)]}): the public inputs ARE the expected outputs, computed with the gadget's own values —
    the caller checks them against the oracle's permutation."""
    rng = np.random.default_rng(seed)
    rnd = lambda: int.from_bytes(rng.bytes(32), "big") % R
    leaves = [rnd() for _ in range(paths)]
    sibs = [[rnd() for _ in range(depth)] for _ in range(paths)]
    blocks = [[rnd() for _ in range(12)] for _ in range(wide)]
    # pass 1 with placeholder public inputs to learn the outputs, pass 2 with them in place (the builder computes every value as it goes)
    pub = [0] * (paths + wide)
    for _ in range(2):
        b = Builder(pub, leaves + [s for p in sibs for s in p] + [x for blk in blocks for x in blk])
        base = b.n_public
        rec = {3: [], 13: []}
        outs = []
        for p in range(paths):
            cur = b.wire(base + p)
            for d in range(depth):
                sib = b.wire(base + paths + p * depth + d)
                st = poseidon_gadget(b, [b.const(0), cur, sib] if (leaves[p] >> d) & 1 else [b.const(0), sib, cur], *params[3], rec[3])
                cur = st[1]
            outs.append(cur)
        for w in range(wide):
            st = poseidon_gadget(b, [b.const(0)] + [b.wire(base + paths + paths * depth + 12 * w + i) for i in range(12)], *params[13], rec[13])
            outs.append(st[1])
        for i, o in enumerate(outs):
            b.assert_mul(o, b.const(1), b.wire(1 + i), "root")
        pub = [b.eval(o) for o in outs]
    return b, rec
