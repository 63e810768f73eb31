#!/usr/bin/env python3
"""Writes tests/golden/collateral_tier_cases.json: the 21-row table of the reference's
circuit/get_and_check_tier_ratios_query_results_test.go:145-170 (TestGetAndCheckTierRatiosQueryResultsEdgeCases) as DATA —
tier lists (:112-131), collateral, tier index, flag, price and whether the reference expects the (index, flag) claim to be
rejected.  No code of the reference is kept: only the literals of its table.  MAX = utils.MaxTierBoundaryValue = 2^118."""
import json
import os

MAX = 1 << 118
std = [[100, 100], [200, 80], [300, 50]]
single80 = [[100, 80]]
floor = [[100, 100], [200, 33]]
zero_ratio = [[100, 100], [200, 0]]
zero_width = [[100, 100], [100, 80], [200, 50]]
rows = [
    ("first_tier_normal_range", std, 60, 0, 0, False),
    ("first_tier_equal_boundary", std, 100, 0, 0, False),
    ("middle_tier_normal_range", std, 150, 1, 0, False),
    ("middle_tier_equal_boundary", std, 200, 1, 0, False),
    ("last_tier_flag_zero", std, 250, 2, 0, False),
    ("flag_one_saturates_to_last_precomputed", std, 350, 2, 1, False),
    ("flag_one_with_equal_last_boundary_should_fail", std, 300, 2, 1, True),
    ("flag_one_with_non_last_index_should_fail", std, 350, 1, 1, True),
    ("index_greater_than_max_should_fail", std, 200, 3, 0, True),
    ("flag_non_boolean_should_fail", std, 150, 1, 2, True),
    ("zero_collateral_index_zero_should_pass", std, 0, 0, 0, False),
    ("zero_collateral_with_index_gt_zero_should_fail", std, 0, 1, 1, True),
    ("index_too_low_for_value_should_fail", std, 250, 1, 0, True),
    ("index_too_high_for_value_should_fail", std, 50, 2, 0, True),
    ("flag_one_value_exceeds_max_tier_boundary_should_fail", std, MAX + 1, 2, 1, True),
    ("single_tier_flag_zero", single80, 70, 0, 0, False),
    ("single_tier_flag_one", single80, 150, 0, 1, False),
    ("single_tier_flag_one_equal_boundary_should_fail", single80, 100, 0, 1, True),
    ("floor_semantics_non_divisible", floor, 150, 1, 0, False),
    ("zero_ratio_tier_increment", zero_ratio, 150, 1, 0, False),
    ("zero_width_tier_equal_boundary", zero_width, 100, 0, 0, False),
]
out = {"source": "circuit/get_and_check_tier_ratios_query_results_test.go:112-170", "max_tier_boundary": str(MAX), "price": 1,
       "cases": [{"name": n, "tiers": t, "collateral": str(c), "index": i, "flag": f, "expect_fail": x} for n, t, c, i, f, x in rows]}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "collateral_tier_cases.json"), "w"), indent=1)
print(len(rows), "cases")
