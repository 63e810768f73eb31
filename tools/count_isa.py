#!/usr/bin/env python3
"""Count the VALU instructions of the steady-state loop of each kernel in a HIP source (gfx950 ISA via hipcc -S): the currency of
this code base is instructions per field product (every candidate multiply instruction issues at the same rate on gfx950,
profiles/r01_alu_microbench.txt).  Usage: tools/count_isa.py tools/fe52_bench.hip [products-per-loop-iteration]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    per_iter = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd", "csrc"),
                               "--cuda-device-only", "-S", src, "-o", out])
        s = open(out).read()
    for m in re.finditer(r"\n(_Z\w+):\s*; @", s):
        name = m.group(1)
        body = s[m.end():s.index("s_endpgm", m.end())]
        blocks = re.split(r"\n(\.LBB\d+_\d+):", body)
        for i in range(1, len(blocks), 2):
            lab, txt = blocks[i], blocks[i + 1]
            if not re.search(r"s_cbranch_\w+ " + re.escape(lab) + r"\b", txt):
                continue
            ins = [l.strip().split()[0] for l in txt.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
            c = collections.Counter(ins)
            valu = sum(v for k, v in c.items() if k.startswith("v_"))
            print(f"{name} loop {lab}: {len(ins)} instructions, {valu} VALU, {valu / per_iter:.1f} VALU per product")
            print("   " + ", ".join(f"{k} {v}" for k, v in c.most_common(12)))


if __name__ == "__main__":
    main()
