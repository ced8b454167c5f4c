"""bench.py's JSON line through its own command line: two ranks on one GPU, the side configs and self-checks, the timed region as a graph.
(Round 6: regrouped by component from the by-round files of rounds 2-5; no test body changed.)"""
import json  # noqa: F401
import os  # noqa: F401
import subprocess  # noqa: F401
import sys  # noqa: F401

import numpy as np  # noqa: F401
import pytest

from oracle import gridstep_oracle as O  # noqa: F401
from sorrel_amd import _native as N  # noqa: F401
from tests import helpers as H  # noqa: F401
from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_run_the_product_and_agree_with_one_process(built):
    """The N>1 control path of bench.py (process group, global-env-id sharding, barrier, max-over-ranks timing, the one
    all-reduce) with the PRODUCT engine in every rank: two ranks on this one GPU (SGW_BENCH_REHEARSAL: gloo collectives)
    must report the same rollout as one process over the same 8 192 global envs.  This test makes no GPU call itself."""
    common = ["--steps", "8", "--warmup", "2", "--prewarm-steps", "0", "--no-cpu-baseline", "--no-series"]
    two = _run_bench(["--gpus", "2", "--envs", "4096"] + common, 2, {"SGW_BENCH_REHEARSAL": "1"})
    one = _run_bench(["--gpus", "1", "--envs", "8192"] + common, 1, {})
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["global_envs"] == one["config"]["global_envs"] == 8192
    assert two["rollout"]["envs"] == one["rollout"]["envs"] == 8192.0
    assert two["rollout"]["sum_total_reward"] == one["rollout"]["sum_total_reward"]
    assert two["rollout"]["status"] == 0 and two["value"] > 0 and two["scaling"] == "weak"
    # global env ids: rank r owns [r * 4096, (r + 1) * 4096)
    assert two["rollout"]["first_env_id_rank0"] == 0 and two["rollout"]["first_env_id_last_rank"] == 4096
    assert one["rollout"]["first_env_id_last_rank"] == 0
    # N > 1 lines carry the roofline (priced on the slowest rank's kernel) and what the closing barrier costs
    for line in (two, one):
        rf = line["roofline"]
        assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0.0 < rf["frac"] < 1.5
        assert rf["kernel_ms"] >= rf["kernel_ms_rank0"] > 0.0
        assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
        assert line["timing"]["barrier_plus_synchronize_ms"] >= 0.0
    assert "configs" not in two and "cpu_baseline" not in two
    # the plain invocation (the shape of the driver's N = 1 command): bench.py starts its ranks itself, as a child process, before any GPU call
    own = _run_bench(["--gpus", "2", "--envs", "4096"] + common, 2, {"SGW_BENCH_REHEARSAL": "1"}, plain=True)
    assert own["n_gpus"] == 2 and own["config"]["process_group_world_size"] == 2 and own["config"]["global_envs"] == 8192
    assert own["rollout"]["sum_total_reward"] == one["rollout"]["sum_total_reward"] and own["rollout"]["first_env_id_last_rank"] == 4096
    assert "all-reduce" in own["config"]["sharding"] and "MAX" in own["config"]["sharding"]


def test_bench_line_carries_side_configs_cold_start_and_the_oracle_self_check(built):
    """The driver's line (N = 1, config 3) with short settings: `configs` holds config 2, config 5's per-GPU share and
    config 3 at 524 288 envs, each with kernel_ms / roofline / kernel; `prewarm_series` shows the cold start; the envs the
    CPU baseline played are replayed on the GPU and must be equal (rollout.checked_vs_oracle)."""
    line = _run_bench(["--gpus", "1", "--steps", "10", "--warmup", "2", "--prewarm-steps", "120", "--side-steps", "10",
                       "--cpu-seconds", "2", "--turns-per-launch", "5"], 1, {})
    assert set(line["configs"]) == {"c2", "c5", "c3_524288"}
    for name, c in line["configs"].items():
        assert c["kernel_ms"] > 0 and 0.0 < c["roofline"]["frac"] < 1.5 and c["status"] == 0, name
        assert "step_" in c["kernel"], name
    assert line["configs"]["c3_524288"]["envs"] == 524288 and line["configs"]["c5"]["envs"] == 2048
    side = line["roofline"]["side_configs"]            # the same, in brief, inside `roofline` (what the driver's record keeps)
    assert set(side) == {"c2", "c5", "c3_524288", "policy_turns_us"}
    turns = side.pop("policy_turns_us")              # wall time of Environment.take_turn through the Python API, and that every variant ends in the same state
    small, many = turns["c3_shape_1024_envs"], turns["c5_shape_2048_envs"]
    assert "error" not in small and "error" not in many, (small, many)
    assert small["recorded_equals_eager"] is True and small["generic_equals_eager"] is True and many["speculative_equals_eager"] is True
    assert min(small["recorded"], small["eager_loop"], small["eager_generic_loop"], many["speculative"], many["eager_loop"]) > 0   # (recorded values; which is faster is the cards' business)
    for name, c in side.items():
        assert c["kernel_ms"] == line["configs"][name]["kernel_ms"] and c["frac"] == line["configs"][name]["roofline"]["frac"]
    assert side["c2"]["checked_vs_oracle_equal"] is True and side["c5"]["checked_vs_oracle_equal"] is True
    pw = line["roofline"]["prewarm_series"]
    assert pw["n"] == 120 and pw["launches_0_10_mean_ms"] > 0 and pw["launches_10_100_mean_ms"] > 0
    chk = line["rollout"]["checked_vs_oracle"]
    assert chk["equal"] is True and chk["tensors_that_differ"] == [] and chk["envs"] == 32768 and chk["turns"] >= 5
    assert chk["sum_total_reward"] == chk["oracle_sum_total_reward"]
    assert "wg_per_cu=8" in line["roofline"]["kernel"] and "cap=auto:0" in line["roofline"]["kernel"]     # what config 3 really launches
    wp = line["roofline"]["write_only_probe"]      # what fill_ reaches on this card: context for frac
    assert wp["bytes"] == 65536 * 8 * 6 * 49 * 4 and 0.5 < wp["tb_per_s"] < 12.0
    pt = line["policy_turn"]                       # the policy-driven turn on the headline's engine
    assert pt["status"] == 0 and pt["launches"] == 9
    assert min(pt["fused_turn_ms"], pt["policy_turn_ms"], pt["policy_turn_replay_rows_ms"], pt["policy_turn_replay_rows_one_launch_ms"]) > 0.0
    # what the driver's record keeps: scalars of `roofline` itself (round 6)
    rf = line["roofline"]
    for name in ("c2", "c5", "c3_524288"):
        assert rf[f"{name}_kernel_ms"] == line["configs"][name]["kernel_ms"] and rf[f"{name}_frac"] == line["configs"][name]["roofline"]["frac"]
    assert rf["c2_checked_vs_oracle_equal"] is True and rf["c5_checked_vs_oracle_equal"] is True
    assert rf["timed_engine_checked_equal"] is True and "step_fast<true, 2, 6, 3, 32, 32>" in rf["timed_engine_checked_what"]
    tchk = line["rollout"]["timed_engine_checked_vs_oracle"]
    assert tchk["envs"] == 65536 and tchk["turns"] == 3 and tchk["tensors_that_differ"] == [] and tchk["from_turn"] >= 120 + 2 + 10
    assert rf["write_only_tb_per_s"] == wp["tb_per_s"] and rf["prewarm_10_100_ms"] == pw["launches_10_100_mean_ms"]
    assert rf["c5_first_placement_frac"] > 0 and rf["c5_median_placement_frac"] > 0 and rf["c5_frac"] >= 0.98 * rf["c5_first_placement_frac"]
    assert rf["policy_turn_replay_rows_one_launch_ms"] == pt["policy_turn_replay_rows_one_launch_ms"] and rf["fused_turn_ms"] == pt["fused_turn_ms"]
    assert rf["c2_rollout_frac"] > 0 and rf["c2_rollout_ms_per_turn"] > 0 and rf["c2_rollout_checked_vs_oracle_equal"] is True and rf["c3_rollout_ms_per_turn"] > 0
    assert rf["take_turn_1024_envs_recorded_equals_eager"] is True and rf["take_turn_c5_speculative_equals_eager"] is True


# ------------------------------------------------------------------ bench.py: the timed launches as one hipGraph
@pytest.mark.gpu
def test_bench_timed_region_as_a_graph_plays_the_same_rollout(built):
    """bench.py --graph captures its K timed sgw_step launches in one hipGraph (each node with the turn number it carries when the
    region runs) and replays it inside the region.  The same launches with the same turn numbers: the rollout it reports -- float64
    totals summed over the batch -- must be exactly what the plain loop reports, with and without pre-warm / re-warm launches in
    front.  This test makes no GPU call itself."""
    from tests.gpu_common import _run_bench

    for extra in (["--prewarm-steps", "0"], ["--prewarm-steps", "40", "--rewarm-steps", "7"]):
        common = ["--gpus", "1", "--envs", "2048", "--steps", "9", "--warmup", "3", "--no-cpu-baseline", "--no-side-configs", "--no-self-check"] + extra
        a = _run_bench(common + ["--graph"], 1, {})
        b = _run_bench(common, 1, {})
        assert "hipGraph" in a["timed_region_submission"] and b["timed_region_submission"] == "K sgw_step calls"
        assert a["rollout"]["sum_total_reward"] == b["rollout"]["sum_total_reward"] and a["rollout"]["status"] == b["rollout"]["status"] == 0
        assert a["steps"] == b["steps"] == 9 and a["roofline"]["kernel_ms"] > 0
