#!/usr/bin/env python3
"""Static VALU instruction mix of the hot kernels by ISSUE CLASS (classes and their measured issue cost: profiles/r03_valu_class.txt,
tools/valu_class_bench.hip).  Usage: tools/valu_mix.py file.s kernel_substring [...]  (file.s from hipcc -S --cuda-device-only)"""
import collections
import re
import sys

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_ashrrev_i32", "v_lshrrev_b32", "v_mov_b32", "v_not_b32",
        "v_add_f32", "v_mul_f32", "v_fma_f32", "v_max_u32", "v_min_u32", "v_max_i32", "v_min_i32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}


def classify(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith("v_cndmask"):
        return "cndmask_vcc" if op.endswith("_e32") else "cndmask_sgpr"
    if base.startswith(("v_mad_i64", "v_mad_u64")):
        return "mad64"
    if base.startswith(("v_mul_lo", "v_mul_hi", "v_mad_", "v_mul_u32_u24", "v_mul_i32_i24")):
        return "mul32"
    if op.endswith("_dpp"):
        return "dpp"
    if base in FAST:
        return "fast32" if not op.endswith("_e64") else "fast32_e64"
    if base.startswith(("v_ashrrev_i64", "v_lshlrev_b64", "v_lshrrev_b64", "v_lshl_add_u64")):
        return "shift64"
    if base.startswith("v_cmp"):
        return "cmp"
    if base.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if base.startswith("v_"):
        return "other_valu(" + base + ")"
    return None


def main():
    s = open(sys.argv[1]).read()
    for want in sys.argv[2:]:
        for m in re.finditer(r"\n(_Z\w+):\s*; @", s):
            name = m.group(1)
            if want not in name:
                continue
            body = s[m.end():s.index("s_endpgm", m.end())]
            ins = [l.strip().split()[0] for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
            c = collections.Counter()
            other = collections.Counter()
            for op in ins:
                k = classify(op)
                if k is None:
                    other[op.split("_")[0] + "_" + (op.split("_")[1] if "_" in op else "")] += 1
                    continue
                if k.startswith("other_valu"):
                    c["other_valu"] += 1; other[k] += 1
                else:
                    c[k] += 1
            valu = sum(c.values())
            print(f"{name[:90]}\n   {len(ins)} instructions, {valu} VALU: " + ", ".join(f"{k} {v} ({100.0 * v / valu:.1f}%)" for k, v in c.most_common()))
            print("   other: " + ", ".join(f"{k} {v}" for k, v in other.most_common(14)))


if __name__ == "__main__":
    main()
