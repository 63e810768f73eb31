#!/usr/bin/env python3
"""Where does a proof's wall time go?  Reads rocprofv3 --kernel-trace (+ optional --hip-trace) CSVs of `bench.py --timed-only` and
prints, for the LAST proof in the trace: per stream the busy time, the union of busy time over all streams, the gaps (no kernel on
any stream) with the kernels either side, and the phases by kernel family.  Usage: tools/timeline_summary.py kernel_trace.csv [hip_api_trace.csv]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_\w+|radix\w*|DeviceRadix\w*|onesweep\w*|histogram\w*)", name)
    base = m.group(1) if m else name[:40]
    if "Fp2" in name or "g2pair" in name or "FpParams>, zk::Fe<zk::FpParams> >" in name:
        pass
    return base


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Stream_Id"]), r["Kernel_Name"]))
    rows.sort()
    # a proof starts with the three k_ntt_pass29 launches... robust marker: the first k_h_pointwise of each proof; take windows between
    # consecutive copies that precede a proof: bench.py copies a0,b0,c0 -> a,b,c (memcpy, not a kernel), so use k_h_pointwise as anchor
    anchors = [i for i, r in enumerate(rows) if "k_h_pointwise" in r[3] or "k_ntt_top29" in r[3]]   # one per proof (the fused top pass since round 3)
    if len(anchors) < 2:
        print("need at least two proofs in the trace"); return
    # proof window: from the first kernel after the previous proof's last kernel... approximate by anchor-to-anchor period
    a0, a1 = anchors[-2], anchors[-1]
    t0, t1 = rows[a0][0], rows[a1][0]
    period = (t1 - t0) / 1e6
    win = [r for r in rows if t0 <= r[0] < t1]
    print(f"one proof period (pointwise step to pointwise step): {period:.2f} ms, {len(win)} kernel launches")
    per_stream = defaultdict(float)
    fam = defaultdict(lambda: [0.0, 0])
    for s, e, st, n in win:
        per_stream[st] += (e - s) / 1e6
        k = short(n)
        if k.startswith("k_acc_level1") or k.startswith("k_acc_levelN") or k.startswith("k_reduce"):
            k += "<G2>" if "Fp2" in n or "g2" in n.lower() else "<G1>"
        fam[k][0] += (e - s) / 1e6; fam[k][1] += 1
    for st, ms in sorted(per_stream.items()):
        print(f"  stream {st}: kernels busy {ms:.2f} ms")
    # union of busy intervals
    iv = sorted((s, e) for s, e, _, _ in win)
    busy = 0; cur_s, cur_e = iv[0]
    gaps = []
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((cur_e, s))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    idle = sum(g[1] - g[0] for g in gaps)
    print(f"  any-stream busy {busy / 1e6:.2f} ms, idle (no kernel running anywhere) {idle / 1e6:.2f} ms in {len(gaps)} gaps")
    gaps.sort(key=lambda g: g[0] - g[1])
    print("  largest gaps (ms, kernel before -> kernel after):")
    for g in gaps[:12]:
        before = max((r for r in win if r[1] <= g[0]), key=lambda r: r[1], default=None)
        after = min((r for r in rows if r[0] >= g[1]), key=lambda r: r[0], default=None)
        print(f"    {(g[1] - g[0]) / 1e6:7.3f}  {short(before[3]) if before else '-'} -> {short(after[3]) if after else '-'}")
    print("  kernel families over the period (ms, launches):")
    for k, (ms, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f"    {k:34s} {ms:8.2f} {c:5d}")
    small = [(e - s) / 1e3 for s, e, _, _ in win if (e - s) < 200e3]
    print(f"  launches shorter than 0.2 ms: {len(small)}, total {sum(small) / 1e3:.2f} ms")
    if len(sys.argv) > 2:
        api = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(sys.argv[2])):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if t0 <= s < t1:
                api[r["Function"]][0] += (e - s) / 1e6; api[r["Function"]][1] += 1
        print("  HIP API calls of the period (host ms, calls):")
        for k, (ms, c) in sorted(api.items(), key=lambda kv: -kv[1][0])[:12]:
            print(f"    {k:34s} {ms:8.2f} {c:6d}")


if __name__ == "__main__":
    main()
