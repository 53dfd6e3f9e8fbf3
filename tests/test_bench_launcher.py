"""`python bench.py --gpus N` must produce an N-rank line by itself (no launcher around it): it re-execs under
torch.distributed.run, one process per rank.  On this GPU-less box the rank plumbing is exercised with --dry-run
(gloo, no world, no kernel): ranks, barriers, max-over-ranks timing, the logging all-gather in global arena order."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3", "--warmup", "1", "--arenas", "8",
                        "--log-every", "2"] + extra, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout   # exactly one JSON line, printed by rank 0
    return json.loads(lines[0])


def test_gpus_2_self_spawns_two_ranks():
    line = _run(["--gpus", "2"])
    assert line["n_gpus"] == 2 and line["dry_run"] is True
    assert len(line["per_rank_env_steps_per_s"]) == 2
    assert line["gathered_rows"] == 16                      # both ranks' blocks arrived in the all-gather
    assert line["config"]["env_steps_per_step"] == 8 * 500 * 2
    assert line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"


def test_two_rank_line_carries_configs4():
    """with more than one rank the line keeps configs[1] as its value and adds extra.configs4 (8192 arenas x 3-vs-3 per rank): here the dry
    run of that branch — both ranks' [8192, 3] blocks arrive in the all-gather, in global arena order"""
    line = _run(["--gpus", "2"])
    c4 = line["extra"]["configs4"]
    assert c4["dry_run"] is True and c4["n_gpus"] == 2 and c4["arenas_per_gpu"] == 8192
    assert c4["gathered_rows"] == 2 * 8192
    assert c4["first_global_arena_of_each_block"] == [0.0, 8192.0]
    assert line["metric"].startswith("env-steps/sec") and line["n_gpus"] == 2   # the headline is still configs[1]


def test_gpus_8_dry_run_is_the_scale_table_the_driver_will_fill():
    """8-rank readiness without hardware: `bench.py --gpus 8` self-spawns eight ranks (gloo here, RCCL on the
    node), weak scaling, per-rank rates, the configs[4] branch with 8 x 8192 rows gathered in global arena order — and, given the 1-GPU line's
    value, the line carries scaling_efficiency = value_8 / (8 x value_1) so that the first real SCALE run yields the table with no edits"""
    line = _run(["--gpus", "8", "--one-gpu-value", "1000.0"])
    assert line["n_gpus"] == 8 and line["dry_run"] is True and line["scaling"] == "weak"
    assert len(line["per_rank_env_steps_per_s"]) == 8 and all(v > 0 for v in line["per_rank_env_steps_per_s"])
    assert line["gathered_rows"] == 8 * 8
    # the keys the REAL N-rank line carries too (bench.py: main_low -> ShardedWorld.evidence): what the collective itself saw
    assert line["ranks_seen"] == 8 and line["backend"] == "gloo"
    assert line["first_global_arena_of_each_block"] == [8 * r for r in range(8)] and line["arenas_of_each_block"] == [8] * 8
    assert line["config"]["env_steps_per_step"] == 8 * 500 * 8
    assert abs(line["scaling_efficiency"] - line["value"] / (8 * 1000.0)) < 1e-12 and line["one_gpu_value"] == 1000.0
    c4 = line["extra"]["configs4"]
    assert c4["dry_run"] is True and c4["n_gpus"] == 8 and c4["arenas_per_gpu"] == 8192
    assert c4["gathered_rows"] == 8 * 8192 and c4["ranks_seen"] == 8                         # BASELINE configs[4]: 65536 arenas over 8 ranks
    assert c4["first_global_arena_of_each_block"] == [float(8192 * r) for r in range(8)]    # blocks in global arena order
    assert "scaling_efficiency" not in _run(["--gpus", "2"])                                 # only when a 1-GPU value was supplied


def test_single_rank_line_contract():
    line = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in line
    assert line["n_gpus"] == 1 and line["vs_baseline"] is None and line["dtype"] == "f64"
