#!/usr/bin/env python3
"""profiles/rNN_pmc_valu.json from one rocprofv3 pass (--pmc SQ_INSTS_VALU --kernel-trace, csv): VALU wave-instructions per
kernel, the input of bench.py's `roofline.valu_issue` (issue-rate ceiling = instructions x 4 cycles / (1024 SIMDs x 2.4 GHz)).
usage: pmc_valu_summary.py counter_collection.csv out.json [round-tag]"""
import collections
import csv
import json
import sys

KERNELS = ("k_acc_level1_fp29", "k_acc_level1_g2pair29", "k_ntt_pass29", "k_ntt_mid29", "k_ntt_top29", "k_acc_levelN29", "k_reduce_level29", "k_reduce_scan29", "k_h_pointwise",
           "k_dsort_count0", "k_dsort_scatter0", "k_dsort_count", "k_dsort_scatter", "k_scan_", "k_r1cs_eval", "k_solve_level", "k_count_queries", "k_decompose", "k_filter_write", "k_filter_count",
           "k_hash2_level", "k_tree_level", "k_account_leaves_coop", "k_account_leaves", "k_cex_commitments_coop", "k_cex_commitments")


PER_PROOF = {"k_acc_level1_fp29": 6}   # launches of one proof: 2 Pedersen sums (tiny) + A, B1, K, Z; the set-up solve adds 2 tiny ones in front of the first proof


def main():
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        if r["Counter_Name"] != "SQ_INSTS_VALU":
            continue
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                rows[k].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r["Grid_Size"])))
                break
    tag = sys.argv[3] if len(sys.argv) > 3 else "r02"
    cmd = sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 1 --warmup 0 --timed-only"
    out = {"source": f"rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -- {cmd} ({tag})",
           "simds": 1024, "issue_cycles_per_wave_instruction": 4, "nominal_clock_hz": 2.4e9, "kernels": {}}
    for k, lst in rows.items():
        lst.sort()
        dropped = 0
        if k in PER_PROOF:      # whole proofs only: VERDICT r04 weak #6 (8 launches averaged where the line's launch time averages the per-proof 6)
            dropped = len(lst) % PER_PROOF[k]
            lst = lst[dropped:]
        v = sum(x[1] for x in lst); n = len(lst); ms = sum(x[2] for x in lst)
        bound_ms = v * 4 / (1024 * 2.4e9) * 1e3
        out["kernels"][k] = {"launches": n, "valu_wave_insts_total": v, "time_ms_total_under_pmc": ms,
                             "issue_bound_ms_total": bound_ms, "frac_of_issue_bound_under_pmc": bound_ms / ms if ms else None}
        if k in PER_PROOF:
            out["kernels"][k]["launches_dropped_in_front"] = dropped
            out["kernels"][k]["per_launch"] = [{"grid": g, "valu_wave_insts": c, "ms_under_pmc": t} for _, c, t, g in lst[-PER_PROOF[k]:]]
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
