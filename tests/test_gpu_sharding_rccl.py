"""SURVEY.md 8e on the real collective backend: torch.distributed "nccl" (= RCCL on ROCm) under torch.distributed.run.  The GPU box
has one MI355X, so the job has one rank; the same script and the same bench.py command run with N ranks on an N-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_args, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1", NCCL_DEBUG="WARN")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)


def test_side_stream_all_gather_equals_the_packed_statistics_blocks():
    p = _torchrun([os.path.join(ROOT, "tools", "rccl_check.py")], 29731)
    assert p.returncode == 0 and "RCCL_CHECK_OK 1 512" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_rccl_log_reports_every_launched_rank():
    """tools/rccl_check.py --launch N starts the job with NCCL_DEBUG=INFO and fails unless RCCL's own log shows N ranks in a communicator of
    N (one rank on this box; `--launch 8` on an 8-GPU node)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_check.py"), "--launch", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0 and "RCCL_LOG_OK ranks=[0] nranks=[1]" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_bench_runs_under_the_drivers_launcher_with_rccl():
    """the driver's multi-GPU command shape with one rank: RANK / WORLD_SIZE from the launcher, backend nccl, barrier +
    max-over-ranks timing, logging all-gather on the side stream; rank 0 prints exactly one JSON line"""
    p = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--arenas", "2048", "--chunk", "50", "--log-every", "2",
                   "--no-cpu-baseline", "--no-extra"], 29733)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 1e6 and len(d["per_rank_env_steps_per_s"]) == 1 and d["scaling"] == "weak"
    # the REAL line says what the collective saw (not only the dry run's)
    assert d["ranks_seen"] == 1 and d["gathered_rows"] == 2048 and d["first_global_arena_of_each_block"] == [0] and d["backend"] == "nccl"
