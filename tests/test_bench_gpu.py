"""-m gpu: bench.py end to end at a small domain (2^18) — the contract of its one JSON line: the metric fields, every timed proof
verified against the trapdoor (both timed regions), the uniform rate, roofline, the host-pointer boundary leg with its own check,
CPU baseline, solver budget, acceptance.  Guards the line the driver records against regressions of any leg."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line on stdout, whatever the native libraries print
    return json.loads(lines[0])


def test_bench_line_small_domain():
    d = _bench("--log2", "18", "--steps", "3", "--warmup", "1", "--cpu-log2", "14", "--r1cs-terms", "7")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "proofs/s" and d["higher_is_better"] and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "solver is NOT included" in d["config"]["workload"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["two_in_flight"]["proofs_in_flight"] == 2 and d["two_in_flight"]["value"] > 0
    o = d["configs"]["zkpor500_200"]     # BASELINE.json configs[2] in the default line: its own timed region, its proofs checked
    assert o["steps"] == 5 and o["value"] > 0 and abs(o["value"] - 1e3 / o["ms_per_step"]) < 1e-6 * o["value"] and o["checked"] == {"proofs": 5, "ok": 5}
    assert d["checked"]["proofs"] == 3 + d["uniform"]["steps"] + d["two_in_flight"]["steps"] + o["steps"] and d["checked"]["ok"] == d["checked"]["proofs"]
    assert d["checked"]["h_verified"] is True and "quotient identity" in d["checked"]["how"]      # h is verified from a, b, c, not trusted
    assert d["checked"]["per_rank_ok_of_total"] == [[d["checked"]["ok"], d["checked"]["proofs"]]]
    assert len(d["per_rank_ms_per_step"]) == 1 and abs(d["per_rank_ms_per_step"][0] - d["ms_per_step"]) < 0.2 * d["ms_per_step"]
    assert d["value_uniform"] == d["uniform"]["value"] and d["value_uniform"] > 0
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["avg_launch_ms"] > 0
    b = d["boundary"]
    assert b["callers"] == 2 and b["checked_ok"] == b["proofs"] and b["value"] > 0 and b["one_caller_ms_per_proof"] > 0
    q = d["r1cs_resident"]     # host w + resident matrices (zkpor_prove_r1cs), its proofs checked as well
    assert q["callers"] == 2 and q["checked_ok"] == q["proofs"] == 11 and q["terms_per_constraint"] == 7 and q["bytes_per_proof"] * 3 < b["bytes_per_proof"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "2^14" in c["sample"]
    assert c["value_uniform_scalars"] > 0 and "witness scalar mixture" in c["sample"]     # the mixture run and the uniform run (no ordering asserted: 2^14 is noise)
    g = d["witness_gen"]     # the device generators, measured (40 users at this size), spot-checked against the oracle
    assert g["checked_against_oracle"] is True and g["users_per_batch"] == 40 and g["accounts_per_s"] > 0 and g["wire_slots_generated"] > 40 * 20000 and g["lookup_results"] == 40 * 29 * 50 and g["integer_divisions"] == 40 * 150
    assert d["acceptance"]["accepted"] == d["acceptance"]["proofs"] == 4
    assert set(d["go_toolchain"]) >= {"go", "version", "module_cache", "note"}      # the probe for the unmodified verifier's toolchain rides in every line
    assert d["solver_budget"]["gpu_ms_per_proof"] == d["ms_per_step"]
    he = d["solver_budget"]["host_executor_measured"]
    assert he["wire_vector_equals_builder"] is True and he["instructions_per_s"]["threads_1"] > 0 and he["hint_calls"] == 6000 * 5
    assert he["instructions_per_s_with_a_b_c"]["threads_1"] > 0
    hr = d["solver_budget"]["host_row_measured"]       # what stays on the host per proof with the solver program on the device
    assert hr["input_values"] == 1 + 5 + 114 * 500 + 1380 * (7 * 50 + 5 * 500 + 30) and hr["host_core_seconds_per_proof"] > 0 and hr["host_cores_per_gpu_at_this_rate"] > 0
    de = d["solver_budget"]["device_executor_measured"]     # the same program on the device (zkpor_solver_*), a wide and a deep shape
    for shape in ("users_side_by_side", "users_chained"):
        assert de[shape]["wire_vector_equals_builder"] is True and de[shape]["instructions_per_s"] > 0 and de[shape]["launches"] <= 2 * de[shape]["levels"]
    assert de["users_side_by_side"]["launches"] == 24 and de["users_chained"]["levels"] > 4000 and de["users_chained"]["launches"] < 20


def test_bench_other_tier_and_timed_only():
    d = _bench("--log2", "17", "--steps", "2", "--warmup", "1", "--config", "zkpor500_200", "--timed-only")
    assert d["config"]["tier"] == "zkpor500_200" and d["config"]["users_per_batch"] == 200
    assert d["checked"] is None and d["value_uniform"] is None and d["two_in_flight"] is None and d["configs"] is None and "boundary" not in d and "cpu_baseline" not in d and "acceptance" not in d


def test_bench_circuit_mode_end_to_end():
    """--circuit T,A,U: the compiled BatchCreateUserCircuit of that shape — the line's workload is then GENERATED (w, a, b, c, committed values from
    the device-solved wires of a synthetic valid batch, a key with the circuit's sparsity) and `end_to_end` times groth16.Prove from the
    assigned inputs: solver program + BSB22 commitment + a, b, c + prove tail, every proof checked"""
    d = _bench("--circuit", "5,20,6", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-boundary", "--e2e-steps", "3")
    cfgd = d["config"]
    assert cfgd["scalars"] == "generated" and "COMPILED BatchCreateUserCircuit" in cfgd["workload"] and "END TO END" in cfgd["workload"]
    assert cfgd["users_per_batch"] == 6 and cfgd["assets_per_user"] == 5
    mix = cfgd["scalar_mix_measured"]
    assert abs(mix["in_{0,1}"] + mix["below_2^16"] + mix["below_2^64"] + mix["wider"] - 1.0) < 1e-3 and mix["in_{0,1}"] > 0.3
    e = d["end_to_end"]
    # the headline IS the end-to-end region: exactly --steps proofs, two workers of the GPU, the tail on its own hardware queues and no compute unit reserved (round 6)
    assert e["steps"] == d["steps"] == 3 and e["value"] == d["value"] == d["end_to_end_value"] and e["ms_per_proof"] == d["ms_per_step"] == d["end_to_end_ms_per_proof"]
    assert abs(e["value"] - 1e3 / e["ms_per_proof"]) < 1e-6 * e["value"] and e["workers_per_gpu"] == 2 and e["tail_reserve_cus"] == 0 and e["tail_mode"] == 1
    assert len(d["per_rank_ms_per_step"]) == 1 and d["what_value_is"].startswith("END TO END")
    up = e["with_input_upload"]                                    # the same with the assigned inputs uploaded from pageable host memory, every proof
    assert up["steps"] == 3 and up["value"] > 0 and up["same_wires"] is True and up["input_bytes_per_proof"] > 32 * 1000 and d["end_to_end_with_input_upload_value"] == up["value"]
    one = e["one_proof_at_a_time"]                                  # one worker, nothing reserved: round 4's shape
    assert one["steps"] == 3 and one["workers_per_gpu"] == 1 and one["tail_reserve_cus"] == 0 and one["same_wires"] is True
    assert e["checked"] == {"proofs": 9, "ok": 9} and e["same_wires_as_headline"] is True and e["constraints_failing_on_device"] == 0
    assert e["next_proofs_hash_chains_prefetched"] is True and e["assertions"].startswith("left out of the run")
    c = e["circuit"]
    assert c["shape_T_A_U"] == [5, 20, 6] and c["constraints"] > 300000 and c["levels"] < 200 and c["committed_wires"] > 40000 and c["census"]["poseidon_perm_t3"] == 28 * 6
    t = d["prove_tail"]                                             # the tail alone, resident inputs: a named sub-figure since round 5
    assert t["steps"] == 3 and t["value"] == d["prove_tail_value"] and abs(t["value"] - 1e3 / t["ms_per_step"]) < 1e-6 * t["value"]
    assert one["value"] < t["value"]                                # the solver costs something
    assert d["checked"]["ok"] == d["checked"]["proofs"] and d["checked"]["proofs"] >= 3 + 9
    for k in ("solve_phase1_ms", "commit_ms", "solve_phase2_ms", "abc_and_prove_tail_ms"):
        assert e["phases_ms_per_proof"][k] >= 0
    rf = d["roofline"]                                              # the launches of the TIMED (end-to-end) region: 6 per proof and worker warm-ups excluded
    assert rf["launches_in_timed_region"] == 6 * 3 and rf["avg_launch_ms"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]


def test_two_ranks_on_one_device_launcher_merge_and_checks():
    """--gpus 2 --share-device: bench.py launches two ranks itself (torch.distributed.run, 127.0.0.1), both prove on device 0 and meet over
    gloo — what the driver's multi-GPU run exercises that a one-GPU box otherwise never does: the launcher, the barrier / MAX contract, the
    per-rank rows of the line, the checks split over the ranks, two keys built side by side"""
    d = _bench("--gpus", "2", "--share-device", "--log2", "17", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-boundary", "--other-config-steps", "2")
    assert d["n_gpus"] == 2 and d["ranks_share_one_device"] is True and d["scaling"] == "weak"
    assert len(d["per_rank_ms_per_step"]) == 2 and len(d["per_rank_key_synth_and_tables_seconds"]) == 2
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]          # whole-job rate: both ranks' proofs over the MAX time
    rows = d["checked"]["per_rank_ok_of_total"]
    assert len(rows) == 2 and all(r[0] == r[1] and r[1] > 0 for r in rows) and d["checked"]["ok"] == d["checked"]["proofs"] == sum(r[1] for r in rows)
    assert d["checked"]["h_verified"] is True and "boundary" not in d and d["two_in_flight"] is None


def test_two_ranks_in_circuit_mode():
    """the default workload (compiled circuit, generated scalars, `end_to_end`) under the multi-rank contract: every rank compiles, builds its key,
    solves and proves; the end-to-end rate is the ranks' proofs over the MAX time; the two-worker region belongs to N = 1 only"""
    d = _bench("--gpus", "2", "--share-device", "--circuit", "5,20,6", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-boundary", "--e2e-steps", "2")
    assert d["n_gpus"] == 2 and d["ranks_share_one_device"] is True and d["config"]["scalars"] == "generated"
    rows = d["checked"]["per_rank_ok_of_total"]
    assert len(rows) == 2 and all(r[0] == r[1] and r[1] > 0 for r in rows) and d["checked"]["ok"] == d["checked"]["proofs"]
    e = d["end_to_end"]
    assert e["steps"] == 2 and e["checked"]["ok"] == e["checked"]["proofs"] == 6 and e["same_wires_as_headline"] is True and e["constraints_failing_on_device"] == 0
    assert abs(e["value"] - 2 * 1e3 / e["ms_per_proof"]) < 1e-6 * e["value"] and e["value"] == d["value"] and len(d["per_rank_ms_per_step"]) == 2
