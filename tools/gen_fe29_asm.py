#!/usr/bin/env python3
"""Generates csrc/fe29_asm.inc: Montgomery product on 9 x 29-bit SIGNED limbs (R = 2^261) as one gfx950 asm statement.
With 29-bit limbs a column of 9+9 partial products (< 2^60 each even for 30-bit 'loose' inputs) fits a 64-bit
accumulator, so the v_addc carry collection of the 8x32 form (half of its instructions) disappears:
162 v_mad_u64_u32 + 18 64-bit shifts + 27 light ops instead of 128 mad + 128 addc + 57.
Limbs are two's-complement int32 (lazy SIGNED residues: differences need no offset, negation is a limb-wise negate):
operand products use v_mad_i64_i32, the column shift is arithmetic (v_ashrrev_i64).  Measured 1.74e11 products/s
(unsigned 29-bit form 1.56e11, 8x32 form 1.25e11).
Contract: |a_i|*|b_j| summed over a column (9 terms, 18 for the fused double product) + 9 * 2^58 must stay below 2^63:
one product: tight x loose (2^29 x 2^30); double product: all operands tight.  Output limbs 0..7 in [0, 2^29), limb 8 signed."""
import os

def gen(two=False):
    L = []
    A = lambda i: f"%[a{i}]"
    B = lambda i: f"%[b{i}]"
    C = lambda i: f"%[c{i}]"
    D = lambda i: f"%[d{i}]"
    Q = lambda i: f"%[q{i}]"
    P = lambda i: f"%[p{i}]"
    acc = "v[0:1]"
    for k in range(18):
        prods = []
        for i in range(max(0, k - 8), min(k, 8) + 1):
            prods.append((A(i), B(k - i)))
            if two:
                prods.append((C(i), D(k - i)))
        for i in range(max(0, k - 8), min(k - 1, 8) + 1):
            if i < k and k - i <= 8:
                prods.append((Q(i), P(k - i)))
        first = (k == 0)
        for (x, y) in prods:
            if first:
                L.append(f"v_mad_i64_i32 {acc}, vcc, {x}, {y}, 0")
                first = False
            else:
                L.append(f"v_mad_i64_i32 {acc}, vcc, {x}, {y}, {acc}")
        if k < 9:
            L.append(f"v_mul_lo_u32 {Q(k)}, v0, %[inv]")
            L.append(f"v_and_b32_e32 {Q(k)}, 0x1fffffff, {Q(k)}")
            L.append(f"v_mad_i64_i32 {acc}, vcc, {Q(k)}, {P(0)}, {acc}")
        elif k < 17:
            L.append(f"v_and_b32_e32 %[r{k - 9}], 0x1fffffff, v0")
        else:
            L.append(f"v_mov_b32_e32 %[r8], v0")   # top limb: everything that is left (result < 2^256)
        if k < 17:
            L.append(f"v_ashrrev_i64 {acc}, 29, {acc}")
    return L

def gen_sqr():
    """a*a: the 36 off-diagonal products are taken once against the doubled limbs d_j = 2 a_j, plus the 9 diagonal ones:
    45 + 9 (doubling) instead of 81 operand products.  Input TIGHT (the doubled limbs must stay below 2^31)."""
    L = []
    A = lambda i: f"%[a{i}]"
    Dd = lambda i: f"%[d{i}]"
    Q = lambda i: f"%[q{i}]"
    P = lambda i: f"%[p{i}]"
    acc = "v[0:1]"
    for i in range(1, 9):
        L.append(f"v_lshlrev_b32_e32 {Dd(i)}, 1, {A(i)}")
    for k in range(18):
        prods = []
        for i in range(max(0, k - 8), min(k, 8) + 1):
            j = k - i
            if i < j:
                prods.append((A(i), Dd(j)))
            elif i == j:
                prods.append((A(i), A(i)))
        for i in range(max(0, k - 8), min(k - 1, 8) + 1):
            if i < k and k - i <= 8:
                prods.append((Q(i), P(k - i)))
        first = (k == 0)
        for (x, y) in prods:
            if first:
                L.append(f"v_mad_i64_i32 {acc}, vcc, {x}, {y}, 0")
                first = False
            else:
                L.append(f"v_mad_i64_i32 {acc}, vcc, {x}, {y}, {acc}")
        if k < 9:
            L.append(f"v_mul_lo_u32 {Q(k)}, v0, %[inv]")
            L.append(f"v_and_b32_e32 {Q(k)}, 0x1fffffff, {Q(k)}")
            L.append(f"v_mad_i64_i32 {acc}, vcc, {Q(k)}, {P(0)}, {acc}")
        elif k < 17:
            L.append(f"v_and_b32_e32 %[r{k - 9}], 0x1fffffff, v0")
        else:
            L.append(f"v_mov_b32_e32 %[r8], v0")
        if k < 17:
            L.append(f"v_ashrrev_i64 {acc}, 29, {acc}")
    return L


def emit_sqr(name):
    lines = gen_sqr()
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    outs = (", ".join(f'[r{i}] "=&v"(r[{i}])' for i in range(9)) + ", " + ", ".join(f'[q{i}] "=&v"(q[{i}])' for i in range(9)) + ", "
            + ", ".join(f'[d{i}] "=&v"(d[{i}])' for i in range(1, 9)))
    ins = ", ".join(f'[a{i}] "v"(a[{i}])' for i in range(9))
    ins += ", " + ", ".join(f'[p{i}] "s"(P::mod29({i}))' for i in range(9)) + ', [inv] "s"(P::INV29)'
    return f'''// r <- a*a*2^-261 mod m; signed limbs, input TIGHT; {len(lines)} instructions (45 operand products instead of 81)
template <class P>
__device__ __forceinline__ void {name}(u32 (&r)[9], const u32 (&a)[9]) {{
    u32 q[9], d[9];
    asm(
{body}
        : {outs}
        : {ins}
        : "vcc", "v0", "v1");
}}
'''


def emit(name, two, ksgpr=False):
    """ksgpr: the first factor of each product (a, and c for the double product) is a wave-uniform CONSTANT held in SGPRs
    (table rows read through the scalar cache): no VGPR copies, 9 or 18 registers less pressure.  VOP3 reads one scalar
    per instruction, which is all a partial product k_i * x_j needs."""
    lines = gen(two)
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    outs = ", ".join(f'[r{i}] "=&v"(r[{i}])' for i in range(9)) + ", " + ", ".join(f'[q{i}] "=&v"(q[{i}])' for i in range(9))
    kc = "s" if ksgpr else "v"
    ins = ", ".join(f'[a{i}] "{kc}"(a[{i}])' for i in range(9)) + ", " + ", ".join(f'[b{i}] "v"(b[{i}])' for i in range(9))
    if two:
        ins += ", " + ", ".join(f'[c{i}] "{kc}"(c[{i}])' for i in range(9)) + ", " + ", ".join(f'[d{i}] "v"(d[{i}])' for i in range(9))
    ins += ", " + ", ".join(f'[p{i}] "s"(P::mod29({i}))' for i in range(9)) + ', [inv] "s"(P::INV29)'
    sig = "u32 (&r)[9], const u32 (&a)[9], const u32 (&b)[9]" + (", const u32 (&c)[9], const u32 (&d)[9]" if two else "")
    doc = ("r <- (a*b + c*d) * 2^-261 mod m; signed limbs, all four inputs TIGHT (|limb| <= 2^29 + 2)" if two
           else "r <- a*b*2^-261 mod m; signed limbs, one input tight, the other |limb| < 2^30")
    if ksgpr:
        doc += "; a" + (" and c" if two else "") + " wave-uniform constants in SGPRs"
    return f'''// {doc}; {len(lines)} instructions, no carry collection, no final subtraction (lazy range)
template <class P>
__device__ __forceinline__ void {name}({sig}) {{
    u32 q[9];
    asm(
{body}
        : {outs}
        : {ins}
        : "vcc", "v0", "v1");
}}
'''


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    txt = "// GENERATED by tools/gen_fe29_asm.py — do not edit.  gfx950 device code only.\n\n" + emit("mont_mul29_asm", False) + "\n" + emit("mont_mul2_29_asm", True) + "\n" + emit_sqr("mont_sqr29_asm") + "\n" + emit("mont_mul29_k_asm", False, True) + "\n" + emit("mont_mul2_29_k_asm", True, True)
    open(os.path.join(here, "..", "zkmerkle-proof-of-solvency_amd", "csrc", "fe29_asm.inc"), "w").write(txt)
    print("mul29:", len(gen(False)), "mul2_29:", len(gen(True)), "sqr29:", len(gen_sqr()))


if __name__ == "__main__":
    main()
