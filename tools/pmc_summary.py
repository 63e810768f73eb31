#!/usr/bin/env python3
"""Builds profiles/rNN_pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE
runs, csv output): HBM bytes per launch for the kernels of interest, with the gfx950 corrections calibrated on kernels of
known traffic (see the `calibration` text).  usage: pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv out.json"""
import collections
import csv
import json
import sys

KERNELS = {  # substring of the kernel name -> (label, FETCH_SIZE correction)
    "k_acc_level1_fp29": ("k_acc_level1_fp29", 1.0),        # 64-byte random gathers are counted 1:1
    "k_acc_level1_g2pair29": ("k_acc_level1_g2pair29", 2.0),
    "k_ntt_pass29": ("k_ntt_pass29", 2.0),                  # wide coalesced streams read back at 1/2 (per launch: see ntt_per_launch below)
    "k_ntt_mid29": ("k_ntt_mid29", 2.0),                    # lowest field: contiguous 8 KiB tiles
    "k_ntt_top29": ("k_ntt_top29", 1.0),                    # highest field: 64-128 B segments, counted 1:1
    "k_dsort_count0": ("k_dsort_count0", 2.0),              # round 6: the digit-stream sort (csrc/sort.hip): coalesced streams in, runs out
    "k_dsort_scatter0": ("k_dsort_scatter0", 2.0),
    "k_dsort_count": ("k_dsort_count", 2.0),
    "k_dsort_scatter": ("k_dsort_scatter", 2.0),
    "k_r1cs_eval": ("k_r1cs_eval", 1.0),
    "k_filter_write": ("k_filter_write", 2.0),
    "k_filter_count": ("k_filter_count", 2.0),
    "k_h_pointwise": ("k_h_pointwise", 2.0),
    "k_acc_levelN29": ("k_acc_levelN29", 2.0),
    "k_reduce_level29": ("k_reduce_level29", 2.0),
}


PER_PROOF = {"k_acc_level1_fp29": 6}   # launches of one proof: 2 Pedersen sums (tiny) + A, B1, K, Z; the set-up solve adds 2 tiny ones in front of the first proof


def load(path, counter):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for sub, (label, _) in KERNELS.items():
            if sub in r["Kernel_Name"]:
                rows[label].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                break
    agg = {}
    for label, lst in rows.items():
        lst.sort()
        if label in PER_PROOF:      # whole proofs only: the launches in front of the first proof belong to the set-up solve
            lst = lst[len(lst) % PER_PROOF[label]:]
        agg[label] = [sum(x[1] for x in lst), len(lst)]
    return agg


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE"); write = load(sys.argv[2], "WRITE_SIZE")
    cmd = sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 1 --warmup 0 --timed-only"
    script = sys.argv[5] if len(sys.argv) > 5 else "the round's profile script under tools/rounds/"
    out = {"command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace --output-format csv -- {cmd} (two separate passes, {script})",
           "calibration": "FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 wide coalesced read streams report exactly 1/2 of the true bytes "
                          "(calibrated in round 1 on k_fr_mul: 4.29 GB true reads -> 2.147 GB reported, k_h_pointwise 6.44 -> 3.22; WRITE_SIZE exact), "
                          "as MI355X_MICROARCH.md says; the 64-byte random point gathers of the G1 level-1 kernel are counted 1:1 "
                          "(expected 64 B point + 8 B key/val = 72 B/entry, reported 73 B/entry).",
           "kernels": {}}
    for _, (label, corr) in KERNELS.items():
        if label not in fetch and label not in write:
            continue
        fb, fn = fetch.get(label, [0.0, 1]); wb, wn = write.get(label, [0.0, 1])
        fpl = fb * 1024 / max(fn, 1); wpl = wb * 1024 / max(wn, 1)
        out["kernels"][label] = {"launches": fn, "fetch_raw_bytes_per_launch": fpl, "fetch_correction": corr,
                                 "write_bytes_per_launch": wpl, "hbm_bytes_per_launch": fpl * corr + wpl}
    # the NTT launches one by one: a launch that reads contiguous tiles (lowest field) reports half of its reads, the others 1:1 —
    # told apart by the raw figure (half an array or less = a halved contiguous stream); total per computeH = the sum over one proof
    arr = float(32 << 26)
    per = collections.defaultdict(lambda: [0.0, 0])
    wr = collections.defaultdict(lambda: [0.0, 0])
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        if r["Counter_Name"] == "FETCH_SIZE" and "k_ntt_" in r["Kernel_Name"]:
            rows.append((r["Dispatch_Id"], r["Kernel_Name"], float(r["Counter_Value"]) * 1024))
    wmap = {}
    for r in csv.DictReader(open(sys.argv[2])):
        if r["Counter_Name"] == "WRITE_SIZE" and "k_ntt_" in r["Kernel_Name"]:
            wmap[r["Dispatch_Id"]] = float(r["Counter_Value"]) * 1024
    total = 0.0
    for did, name, raw in rows:
        corr = 2.0 if raw < 0.75 * arr else 1.0
        total += raw * corr + wmap.get(did, 0.0)
    if rows:
        proofs = max(1, sum(1 for _, name, _ in rows if "k_ntt_top29" in name))      # one fused top pass per computeH: the trace may hold several proofs
        out["ntt_proofs_in_trace"] = proofs
        out["ntt_hbm_bytes_per_computeH"] = total / proofs
        out["ntt_launches_per_computeH"] = len(rows) / proofs
        out["ntt_note"] = ("per launch: FETCH_SIZE x 2 when the raw figure is below 3/4 of the array (contiguous tiles of the lowest field are "
                           "read back at 1/2), x 1 otherwise (strided 64-128 B segments + tabulated twiddles), + WRITE_SIZE; summed over the trace and divided by its proofs (k_ntt_top29 launches)")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
