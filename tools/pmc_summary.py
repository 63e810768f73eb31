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
    "k_ntt_pass29": ("k_ntt_pass29", 2.0),                  # wide coalesced streams read back at 1/2
    "k_h_pointwise": ("k_h_pointwise", 2.0),
    "k_acc_levelN29": ("k_acc_levelN29", 2.0),
    "k_reduce_level29": ("k_reduce_level29", 2.0),
}


def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for sub, (label, _) in KERNELS.items():
            if sub in r["Kernel_Name"]:
                agg[label][0] += float(r["Counter_Value"]); agg[label][1] += 1
                break
    return agg


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE"); write = load(sys.argv[2], "WRITE_SIZE")
    out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --log2 26 --steps 1 "
                      "--warmup 0 --timed-only (two separate passes, tools/r02_profile.sh)",
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
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
