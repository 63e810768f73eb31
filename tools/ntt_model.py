"""Index-exact Python model of the multi-pass LDS NTT in csrc/ntt.hip (field split, in-tile radix-2 stages,
inter-pass twiddles w_N^(l*k*2^s0) from two half-size tables, fused coset / 1/N scaling).  Used to validate the
index algebra against a naive DFT before it is transcribed to HIP (tests/test_ntt_model.py)."""
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
W28 = pow(5, (R - 1) >> 28, R)
G = 5


def rev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def plan_fields(n, kb_low=8, kb_max=9):
    """fields as (lo, kb) from the lowest; DIF processes them high->low, DIT low->high"""
    if n <= kb_low:
        return [(0, n)]
    rest = n - kb_low
    k = (rest + kb_max - 1) // kb_max
    base, extra = divmod(rest, k)
    fields = [(0, kb_low)]
    lo = kb_low
    for i in range(k):
        kb = base + (1 if i < extra else 0)
        fields.append((lo, kb))
        lo += kb
    return fields


class Tables:
    def __init__(self, n, inverse):
        self.n = n
        w = pow(W28, 1 << (28 - n), R)
        if inverse:
            w = pow(w, R - 2, R)
        self.w = w
        self.tb = (n + 1) // 2
        self.lo = [pow(w, e, R) for e in range(1 << self.tb)]
        self.hi = [pow(w, e << self.tb, R) for e in range(1 << (n - self.tb))]
        w512 = pow(W28, 1 << 19, R)
        if inverse:
            w512 = pow(w512, R - 2, R)
        self.small = [pow(w512, j, R) for j in range(256)]

    def tw(self, e):
        e &= (1 << self.n) - 1
        return self.lo[e & ((1 << self.tb) - 1)] * self.hi[e >> self.tb] % R

    def small_tw(self, kb, j):  # w_{2^kb}^j, j < 2^(kb-1)
        return self.small[j << (9 - kb)]


def pass_dif(x, n, lo, kb, T, post_scale=None):
    s0 = n - lo - kb
    for hi in range(1 << s0):
        for l in range(1 << lo):
            base = (hi << (lo + kb)) + l
            v = [x[base + (m << lo)] for m in range(1 << kb)]
            for j in range(kb):
                half = 1 << (kb - 1 - j)
                for q in range(1 << (kb - 1)):
                    grp, pos = divmod(q, half)
                    i0 = grp * 2 * half + pos
                    i1 = i0 + half
                    a, b = v[i0], v[i1]
                    v[i0] = (a + b) % R
                    v[i1] = (a - b) * T.small_tw(kb, pos << j) % R
            for m in range(1 << kb):
                km = rev(m, kb)
                val = v[m]
                if lo > 0:
                    val = val * T.tw((l * km) << s0) % R
                p = base + (m << lo)
                if post_scale is not None:
                    val = val * post_scale(p) % R
                x[p] = val


def pass_dit(x, n, lo, kb, T, pre_scale=None):
    s0 = n - lo - kb
    for hi in range(1 << s0):
        for l in range(1 << lo):
            base = (hi << (lo + kb)) + l
            v = []
            for m in range(1 << kb):
                p = base + (m << lo)
                val = x[p]
                if pre_scale is not None:
                    val = val * pre_scale(p) % R
                if lo > 0:
                    val = val * T.tw((l * rev(m, kb)) << s0) % R
                v.append(val)
            for j in range(kb):
                half = 1 << j
                for q in range(1 << (kb - 1)):
                    grp, pos = divmod(q, half)
                    i0 = grp * 2 * half + pos
                    i1 = i0 + half
                    a, b = v[i0], v[i1] * T.small_tw(kb, pos << (kb - 1 - j)) % R
                    v[i0] = (a + b) % R
                    v[i1] = (a - b) % R
            for m in range(1 << kb):
                x[base + (m << lo)] = v[m]


def fft(x, n, inverse, dif, on_coset, kb_low=8, kb_max=9):
    """gnark-crypto semantics: FFT/FFTInverse with DIF (natural in, bit-reversed out) or DIT (bit-reversed in,
    natural out); coset shift g = 5"""
    x = list(x)
    T = Tables(n, inverse)
    fields = plan_fields(n, kb_low, kb_max)
    ninv = pow(1 << n, R - 2, R)
    ginv = pow(G, R - 2, R)
    if dif:
        if on_coset and not inverse:  # forward DIF on coset: natural input scaled by g^i up front
            x = [v * pow(G, i, R) % R for i, v in enumerate(x)]
        order = list(reversed(fields))
        for idx, (lo, kb) in enumerate(order):
            post = None
            if idx == len(order) - 1 and inverse:
                if on_coset:
                    post = lambda p: pow(ginv, rev(p, n), R) * ninv % R
                else:
                    post = lambda p: ninv
            pass_dif(x, n, lo, kb, T, post)
    else:
        for idx, (lo, kb) in enumerate(fields):
            pre = None
            if idx == 0 and on_coset and not inverse:
                pre = lambda p: pow(G, rev(p, n), R)
            pass_dit(x, n, lo, kb, T, pre)
        if inverse:
            if on_coset:
                x = [v * pow(ginv, i, R) % R * ninv % R for i, v in enumerate(x)]
            else:
                x = [v * ninv % R for v in x]
    return x


def dft_naive(x, n, inverse=False, on_coset=False):
    N = 1 << n
    w = pow(W28, 1 << (28 - n), R)
    if inverse:
        w = pow(w, R - 2, R)
    out = []
    for k in range(N):
        base = pow(w, k, R)
        if on_coset and not inverse:
            base = base * G % R
        acc, p = 0, 1
        for j in range(N):
            acc = (acc + x[j] * p) % R
            p = p * base % R
        out.append(acc)
    if inverse:
        ninv = pow(N, R - 2, R)
        ginv = pow(G, R - 2, R)
        out = [v * ninv % R * (pow(ginv, k, R) if on_coset else 1) % R for k, v in enumerate(out)]
    return out


if __name__ == "__main__":
    import random
    random.seed(1)
    for n, kl, km in [(3, 8, 9), (6, 2, 2), (7, 2, 3), (7, 3, 2), (8, 3, 3), (5, 1, 2)]:
        x = [random.randrange(R) for _ in range(1 << n)]
        for inverse in (False, True):
            for coset in (False, True):
                ref = dft_naive(x, n, inverse, coset)
                got = fft(x, n, inverse, True, coset, kl, km)
                ok1 = [got[rev(k, n)] for k in range(1 << n)] == ref
                xin = [x[rev(i, n)] for i in range(1 << n)]
                got2 = fft(xin, n, inverse, False, coset, kl, km)
                ok2 = got2 == ref
                print(n, kl, km, plan_fields(n, kl, km), "inv" if inverse else "fwd", "coset" if coset else "     ", ok1, ok2)
