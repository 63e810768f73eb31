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


# ---------------------------------------------------------------------------------------------------------------------
# Sharded transform (csrc/ntt.hip ntt_shard_stage): the array is spread over W = 2^wlog ranks.  Every pass over an
# index field is local under one of two distributions of the memory position p:
#   D_low  : rank = low wlog bits of p, local index = p >> wlog            -> all fields above the lowest are local
#   D_high : rank = top wlog bits of p, local index = p mod 2^(n - wlog)   -> all fields below the highest are local
# A pass runs on the local array exactly as on an array of 2^(n-wlog) elements (lo shifted down by wlog under D_low); only the
# exponents of the inter-pass twiddles and of the coset scale need the GLOBAL position: p_glob = ((p_loc << sh) | orv) + add.
def _gmap(n, wlog, rank, dist):
    if dist == "low":
        return wlog, rank, 0
    return 0, 0, rank << (n - wlog)


def pass_local(xloc, n, wlog, rank, dist, lo, kb, T, dif, scale=None):
    """one field pass on a rank's local array; (lo, kb) is the GLOBAL field, T the GLOBAL tables"""
    sh, orv, add = _gmap(n, wlog, rank, dist)
    nl = n - wlog
    lol = lo - sh
    assert (dist == "low" and lol > 0) or (dist == "high" and lo + kb <= nl)
    s0g = n - lo - kb
    for hi in range(1 << (nl - lol - kb)):
        for l in range(1 << lol):
            base = (hi << (lol + kb)) + l
            lg = (l << sh) | orv
            pos = [base + (m << lol) for m in range(1 << kb)]
            pg = [((p << sh) | orv) + add for p in pos]
            v = [xloc[p] for p in pos]
            if not dif:
                for m in range(1 << kb):
                    if scale is not None:
                        v[m] = v[m] * scale(pg[m]) % R
                    if lo > 0:
                        v[m] = v[m] * T.tw((lg * rev(m, kb)) << s0g) % R
            for j in range(kb):
                half = 1 << ((kb - 1 - j) if dif else j)
                for q in range(1 << (kb - 1)):
                    grp, ps = divmod(q, half)
                    i0 = grp * 2 * half + ps
                    i1 = i0 + half
                    if dif:
                        a, b = v[i0], v[i1]
                        v[i0] = (a + b) % R
                        v[i1] = (a - b) * T.small_tw(kb, ps << j) % R
                    else:
                        a, b = v[i0], v[i1] * T.small_tw(kb, ps << (kb - 1 - j)) % R
                        v[i0] = (a + b) % R
                        v[i1] = (a - b) % R
            for m in range(1 << kb):
                val = v[m]
                if dif:
                    if lo > 0:
                        val = val * T.tw((lg * rev(m, kb)) << s0g) % R
                    if scale is not None:
                        val = val * scale(pg[m]) % R
                xloc[pos[m]] = val


def exchange(locs, n, wlog, to_high):
    """all-to-all between the distributions.  low -> high: rank r's local array is already grouped by destination (chunk d =
    local indices [d*M, (d+1)*M)); the receiver interleaves the W chunks (index i*W + r).  high -> low: the sender first
    de-interleaves (chunk d = its elements with local index = d mod W), the receiver concatenates the chunks by sender."""
    W = 1 << wlog
    M = 1 << (n - 2 * wlog)
    out = [[None] * (M * W) for _ in range(W)]
    for s in range(W):
        for d in range(W):
            if to_high:
                chunk = locs[s][d * M:(d + 1) * M]                      # send buffer: contiguous
                for i in range(M):
                    out[d][i * W + s] = chunk[i]                         # receiver-side transpose [W][M] -> [M][W]
            else:
                chunk = [locs[s][i * W + d] for i in range(M)]           # sender-side transpose [M][W] -> [W][M]
                out[d][s * M:(s + 1) * M] = chunk
    return out


def fft_sharded(x, n, wlog, inverse, dif, scale_first=None, scale_last=None, kb_low=8, kb_max=9):
    """the same passes as fft() (without its up-front / trailing host-side scalings: pass them as scale_first / scale_last on
    GLOBAL positions), run rank by rank.  DIF: input in D_low, output in D_high; DIT: input in D_high, output in D_low."""
    W = 1 << wlog
    T = Tables(n, inverse)
    fields = plan_fields(n, kb_low, kb_max)
    assert len(fields) >= 2 and wlog <= fields[0][1] and wlog <= fields[-1][1]
    if dif:
        locs = [[x[(j << wlog) | r] for j in range(1 << (n - wlog))] for r in range(W)]
        steps = list(reversed(fields))
    else:
        locs = [x[r << (n - wlog):(r + 1) << (n - wlog)] for r in range(W)]
        steps = list(fields)
    dist = "low" if dif else "high"
    for idx, (lo, kb) in enumerate(steps):
        need = "high" if lo == 0 else "low"
        if need != dist:
            locs = exchange(locs, n, wlog, to_high=(need == "high"))
            dist = need
        sc = scale_first if idx == 0 else (scale_last if idx == len(steps) - 1 else None)
        for r in range(W):
            pass_local(locs[r], n, wlog, r, dist, lo, kb, T, dif, sc)
    out = [None] * (1 << n)
    for r in range(W):
        for j, v in enumerate(locs[r]):
            out[(r << (n - wlog)) + j if dist == "high" else (j << wlog) | r] = v
    return out


def _selftest_sharded():
    import random
    random.seed(2)
    ok = True
    for n, kl, km, wlog in [(7, 3, 2, 1), (7, 3, 2, 2), (8, 3, 3, 2), (9, 3, 3, 2), (10, 4, 3, 3), (6, 3, 3, 2)]:
        x = [random.randrange(R) for _ in range(1 << n)]
        ginv = pow(G, R - 2, R); ninv = pow(1 << n, R - 2, R)
        for inverse in (False, True):
            # DIF, plain (inverse: 1/N at the last store, as fft() does inside its last pass)
            last = (lambda p: ninv) if inverse else None
            ref = fft(x, n, inverse, True, False, kl, km)
            got = fft_sharded(x, n, wlog, inverse, True, None, last, kl, km)
            ok &= got == ref
            # DIT on the coset (forward: g^rev(p) at the first load)
            if not inverse:
                ref = fft(x, n, False, False, True, kl, km)
                got = fft_sharded(x, n, wlog, False, False, (lambda p: pow(G, rev(p, n), R)), None, kl, km)
                ok &= got == ref
            else:   # inverse DIF on the coset: g^-rev(p)/N at the last store
                ref = fft(x, n, True, True, True, kl, km)
                got = fft_sharded(x, n, wlog, True, True, None, (lambda p: pow(ginv, rev(p, n), R) * ninv % R), kl, km)
                ok &= got == ref
        print("sharded", n, kl, km, "W =", 1 << wlog, ok)
    return ok
