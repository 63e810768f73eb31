"""Poseidon (x^5, BN254 Fr) parameter generation by the published Grain-LFSR procedure
(Grassi et al., "Poseidon", reference script generate_parameters_grain.sage: field=1, sbox=0, n=254,
t, R_F=8, R_P(t)).  The bnb-chain gnark-crypto fork's fr/poseidon package (pinned in the reference's
go.mod:57-60; NOT present under /root/reference) uses these circomlib/iden3 parameter sets.
Validated by tests/test_poseidon_kat.py against the data fixture src/verifier/config/user_config.json.
"""
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
N_BITS = 254
R_F = 8
# partial rounds for width t = 2..17 (iden3/circomlib table)
R_P_TABLE = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]


def r_p(t):
    return R_P_TABLE[t - 2]


class Grain:
    def __init__(self, t, rf, rp, n=N_BITS, field=1, sbox=0):
        bits = []
        def put(v, w):
            bits.extend(int(b) for b in bin(v)[2:].zfill(w))
        put(field, 2); put(sbox, 4); put(n, 12); put(t, 12); put(rf, 10); put(rp, 10)
        bits.extend([1] * 30)
        assert len(bits) == 80
        self.s = bits
        for _ in range(160):
            self._step()

    def _step(self):
        s = self.s
        nb = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(nb)
        return nb

    def bit(self):
        # self-shrinking: take a pair, output 2nd if the 1st is 1, else discard
        while True:
            b1 = self._step()
            b2 = self._step()
            if b1 == 1:
                return b2

    def bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v

    def field_elem_rejection(self):
        while True:
            v = self.bits(N_BITS)
            if v < R:
                return v


def generate(t):
    """returns (round_constants[(R_F+R_P)*t], mds[t][t]) as python ints"""
    rp = r_p(t)
    g = Grain(t, R_F, rp)
    rc = [g.field_elem_rejection() for _ in range((R_F + rp) * t)]
    # Cauchy MDS: M[i][j] = 1/(x_i + y_j); x,y sampled WITHOUT rejection (reduced mod r)
    while True:
        lst = [g.bits(N_BITS) % R for _ in range(2 * t)]
        if len(set(lst)) != 2 * t:
            continue
        xs, ys = lst[:t], lst[t:]
        ok = all((xs[i] + ys[j]) % R != 0 for i in range(t) for j in range(t))
        if ok:
            break
    mds = [[pow((xs[i] + ys[j]) % R, R - 2, R) for j in range(t)] for i in range(t)]
    return rc, mds


def permute(state, rc, mds):
    t = len(state)
    rp = r_p(t)
    k = 0
    for rnd in range(R_F + rp):
        state = [(state[i] + rc[k + i]) % R for i in range(t)]
        k += t
        if rnd < R_F // 2 or rnd >= R_F // 2 + rp:
            state = [pow(x, 5, R) for x in state]
        else:
            state[0] = pow(state[0], 5, R)
        state = [sum(mds[i][j] * state[j] for j in range(t)) % R for i in range(t)]
    return state


_cache = {}


def params(t):
    if t not in _cache:
        _cache[t] = generate(t)
    return _cache[t]


def poseidon(inputs):
    """bnb fork semantics: blocks of 12 chained through state[0]"""
    assert len(inputs) >= 1
    cap = 0
    i = 0
    n = len(inputs)
    while i < n:
        blk = inputs[i:i + 12]
        i += 12
        st = [cap] + [x % R for x in blk]
        rc, mds = params(len(st))
        cap = permute(st, rc, mds)[0]
    return cap


if __name__ == "__main__":
    print(hex(params(3)[0][0]))
    print(hex(params(3)[1][0][0]))
    print(hex(poseidon([1, 2])))


# ---- optimised partial rounds (sparse matrices), the schedule csrc/poseidon.hip uses ------------------------------------
def _mat_inv(a):
    n = len(a)
    m = [row[:] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(a)]
    for c in range(n):
        piv = next(r for r in range(c, n) if m[r][c] % R)
        m[c], m[piv] = m[piv], m[c]
        inv = pow(m[c][c], R - 2, R)
        m[c] = [x * inv % R for x in m[c]]
        for r in range(n):
            if r != c and m[r][c]:
                f = m[r][c]
                m[r] = [(x - f * y) % R for x, y in zip(m[r], m[c])]
    return [row[n:] for row in m]


def optimise(t):
    """factor every partial-round matrix N_i = diag(1, Mhat_i) * [[m00, v],[Mhat_i^-1 w, I]] and push the block-diagonal
    factor through the next round's (element-0-only) S-box: N_(i+1) = M * diag(1, Mhat_i).  Returns per-round
    (constants k_i, m00, v, what) and the trailing block Mhat_last."""
    rc, mds = params(t)
    rp = r_p(t)
    N = [row[:] for row in mds]
    rounds = []
    prev_inv = None
    for i in range(rp):
        mhat = [row[1:] for row in N[1:]]
        w = [N[r][0] for r in range(1, t)]
        v = N[0][1:]
        inv = _mat_inv(mhat)
        what = [sum(inv[r][c] * w[c] for c in range(t - 1)) % R for r in range(t - 1)]
        c = rc[(R_F // 2 + i) * t:(R_F // 2 + i + 1) * t]
        if prev_inv is None:
            k = c[:]
        else:
            k = [c[0]] + [sum(prev_inv[r][cc] * c[1 + cc] for cc in range(t - 1)) % R for r in range(t - 1)]
        rounds.append((k, N[0][0], v, what))
        prev_inv = inv
        last = mhat
        d = [[1 if (r == 0 and cc == 0) else 0 for cc in range(t)] for r in range(t)]
        for r in range(1, t):
            for cc in range(1, t):
                d[r][cc] = mhat[r - 1][cc - 1]
        N = [[sum(mds[r][x] * d[x][cc] for x in range(t)) % R for cc in range(t)] for r in range(t)]
    return rounds, last


def permute_optimised(state):
    t = len(state)
    rc, mds = params(t)
    rp = r_p(t)
    rounds, post = optimise(t)
    s = list(state)
    def full(s, r):
        s = [(s[i] + rc[r * t + i]) % R for i in range(t)]
        s = [pow(x, 5, R) for x in s]
        return [sum(mds[i][j] * s[j] for j in range(t)) % R for i in range(t)]
    for r in range(R_F // 2):
        s = full(s, r)
    for (k, m00, v, what) in rounds:
        s = [(s[i] + k[i]) % R for i in range(t)]
        x0 = pow(s[0], 5, R)
        new0 = (m00 * x0 + sum(v[j] * s[1 + j] for j in range(t - 1))) % R
        s = [new0] + [(s[1 + j] + what[j] * x0) % R for j in range(t - 1)]
    s = [s[0]] + [sum(post[r][c] * s[1 + c] for c in range(t - 1)) % R for r in range(t - 1)]
    for r in range(R_F // 2 + rp, R_F + rp):
        s = full(s, r)
    return s
