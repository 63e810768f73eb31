#!/usr/bin/env python3
"""profiles/rNN_clock.txt from one rocprofv3 pass (--pmc GRBM_GUI_ACTIVE --kernel-trace, csv): the clock a kernel actually ran at
= GRBM_GUI_ACTIVE (GPU-busy cycles, one GRBM instance) / the dispatch's wall time (MI355X_MICROARCH.md, DVFS give-back).
When the pass also carries SQ_BUSY_CYCLES / SQ_INSTS_VALU the same table prices the VALU issue bound at THAT clock.
usage: pmc_clock_summary.py counter_collection.csv out.txt [round-tag]"""
import collections
import csv
import sys

KERNELS = ("k_acc_level1_fp29", "k_acc_level1_g2pair29", "k_ntt_pass29", "k_ntt_mid29", "k_ntt_top29", "k_acc_levelN29", "k_reduce_level29",
           "k_reduce_scan29", "k_dsort_count0", "k_dsort_scatter0", "k_dsort_count", "k_dsort_scatter", "k_decompose", "k_filter_write", "k_solve_level", "k_solve_narrow", "k_r1cs_eval", "k_gadget_poseidon",
           "k_hash2_level", "k_tree_level", "k_account_leaves_coop", "k_account_leaves", "k_cex_commitments_coop", "k_cex_commitments")


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0, 0.0]))   # kernel -> counter -> [sum, launches, ms]
    for r in csv.DictReader(open(sys.argv[1])):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                a = per[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
                a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                break
    tag = sys.argv[3] if len(sys.argv) > 3 else "r04"
    lines = [f"# {tag}: effective clock per kernel = GRBM_GUI_ACTIVE / dispatch wall time (rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace, csv;",
             f"# {sys.argv[4] if len(sys.argv) > 4 else 'python bench.py --steps 1 --warmup 0 --timed-only'}).  GRBM_GUI_ACTIVE is summed over the counter's instances by rocprofv3:",
             "# 'per_instance' divides by the instance count inferred from the longest kernel (a value near 2.4 GHz x its wall time).",
             f"{'kernel':28s} {'launches':>8s} {'ms_total':>10s} {'GRBM_GUI_ACTIVE':>18s} {'cycles/ms':>12s}"]
    rows = []
    for k, cs in per.items():
        if "GRBM_GUI_ACTIVE" not in cs:
            continue
        v, n, ms = cs["GRBM_GUI_ACTIVE"]
        rows.append((k, n, ms, v, v / ms if ms else 0.0))
    # the counter may be reported as the sum over XCDs (8 GRBM instances): infer the divisor from the plausible clock range
    div = 1
    if rows:
        k0 = max(rows, key=lambda r_: r_[2])
        per_ms = k0[4]
        for d in (1, 2, 4, 8, 16, 32):
            if 1.0e6 <= per_ms / d <= 2.6e6:
                div = d
                break
    for k, n, ms, v, cpm in sorted(rows, key=lambda r_: -r_[2]):
        lines.append(f"{k:28s} {n:8d} {ms:10.3f} {v:18.0f} {cpm:12.0f}   -> {cpm / div / 1e6:.3f} GHz (instances: {div})")
    for k, cs in per.items():
        if "SQ_BUSY_CYCLES" in cs:
            v, n, ms = cs["SQ_BUSY_CYCLES"]
            lines.append(f"{k:28s} SQ_BUSY_CYCLES {v:.0f} over {ms:.3f} ms = {v / ms:.0f} per ms")
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
