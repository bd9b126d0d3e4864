"""-m gpu: the N>1 path on real GPUs over RCCL (torch.distributed backend "nccl").  Skipped on a 1-GPU box; the same
engine is covered on CPU by the world_size-2 gloo tests (tests/test_distributed_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")
def test_bench_two_ranks_over_rccl():
    """`python bench.py --gpus 2` spawns its own two ranks (reference run_generation.py:265-266), exchanges the gradients over
    RCCL and reports what it connected."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "opt-125m", "--batch", "4", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["exchange"]["rccl_ranks"] == 2
    assert line["exchange"]["exchange_bytes_per_step_per_gpu"] > 0 and line["config"]["parallelism"] == "dp2"


def test_bench_self_launch_refuses_a_mismatched_world(monkeypatch):
    """WORLD_SIZE from a launcher must agree with --gpus: a silent 1-rank run that prints n_gpus = 1 is exactly what the
    scaling measurement must never get."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
