"""One-off source transformation (kept for the record, idempotent): gives every `int32_t zkpor_*` entry point defined in csrc/*.hip a
function-try-block ending in ZK_ABI_CATCH (common.cuh), so that no C++ exception (std::bad_alloc from a std::vector, std::system_error
from a thread, ...) crosses the C ABI — include/zkpor.h promises "never throw", and a cgo caller cannot unwind (VERDICT r04 weak #1b).

    python tools/abi_firewall.py            # rewrites zkmerkle-proof-of-solvency_amd/csrc/*.hip in place
    python tools/abi_firewall.py --check    # exit 1 if an entry point is not wrapped
"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sorted(glob.glob(os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd", "csrc", "*.hip")))
HEAD = re.compile(r"^int32_t zkpor_\w+\(")
CLOSE = re.compile(r"^\} ZK_ABI_CATCH(_IN\(.*\))?\s*$")      # ZK_ABI_CATCH_IN(ctx): the text goes into that context (common.cuh)


def transform(lines):
    out = list(lines)
    missing = []
    i = 0
    while i < len(out):
        if HEAD.match(out[i]) and not out[i].rstrip().endswith(";"):
            # the line that opens the body: first line from here whose code ends in "{" or that holds the whole body
            j = i
            while "{" not in out[j]:
                j += 1
            line = out[j]
            name = HEAD.match(out[i]).group(0)
            if line.count("{") == line.count("}"):       # one-line body
                if " try {" not in line:
                    missing.append(name)
                    k = line.index("{")
                    out[j] = line[:k] + "try " + line.rstrip("\n") [k:] + " ZK_ABI_CATCH\n"
                i = j + 1
                continue
            if not line.rstrip().endswith("{"):
                raise SystemExit(f"cannot parse the body opener of {name}: {line!r}")
            wrapped = line.rstrip().endswith("try {")
            e = j + 1
            while out[e].rstrip("\n") != "}" and not CLOSE.match(out[e]):
                e += 1
            if not wrapped:
                missing.append(name)
                out[j] = line.rstrip()[:-1] + "try {\n"
                out[e] = "} ZK_ABI_CATCH\n"
            i = e + 1
            continue
        i += 1
    return out, missing


def main():
    check = "--check" in sys.argv
    bad = 0
    for p in SRC:
        with open(p) as f:
            lines = f.readlines()
        new, missing = transform(lines)
        if missing:
            bad += len(missing)
            print(f"{os.path.basename(p)}: {len(missing)} entry points {'NOT wrapped' if check else 'wrapped'}: {', '.join(m[8:-1] for m in missing)}")
            if not check:
                with open(p, "w") as f:
                    f.writelines(new)
    if check and bad:
        sys.exit(1)
    print("ok" if not bad or not check else "")


if __name__ == "__main__":
    main()
