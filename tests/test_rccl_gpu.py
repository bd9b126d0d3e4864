"""-m gpu: the N>1 path on real GPUs: over RCCL (torch.distributed backend "nccl") when the box has >= 2 GPUs, and on ONE GPU with two
ranks sharing it over gloo (device tensors, hooks, async all-reduces, fused AdamW); the same
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


def _run_worker(backend, nproc=2):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MMGL_DIST_BACKEND=backend)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_two_ranks_on_one_gpu_engine_with_device_tensors():
    """The N-rank engine on DEVICE tensors with the hardware a 1-GPU lease has: two ranks share GPU 0, collectives over gloo
    (reference DDP wiring, run_generation.py:283, 317-319, 485).  The worker asserts: exchanged gradient == locally accumulated sum
    of both ranks' gradients; identical bucket launch order on both ranks (several buckets per step); bit-identical parameters on
    both ranks after two fused-AdamW steps although the ranks were constructed with different weights."""
    rep = _run_worker("gloo")
    assert rep["world"] == 2 and rep["params_equal"] and rep["buckets"] >= 3 and rep["exchange_bytes"] > 0
    assert all(s["grad_rel_err"] < 2e-5 for s in rep["steps"])


def test_one_rank_engine_over_rccl_with_the_exchange_forced_on():
    """RCCL on the GPU this box has: a world_size-1 `nccl` process group on cuda:0 with the engine's exchange forced on -- the
    post-accumulate-grad hooks issue `dist.all_reduce(async_op=True)` per bucket (index order) on RCCL's stream during backward,
    `finish_backward` waits on the works, the persistent GEMM runs on its dynamic tile schedule, fused AdamW steps the flat buffers.
    With one rank the sum is the identity: the worker asserts the exchanged gradient is BIT-identical to the same step without the
    exchange, both steps (reference DDP wiring: run_generation.py:283, 317-319, 485)."""
    rep = _run_worker("nccl", nproc=1)
    assert rep["world"] == 1 and rep["backend"] == "nccl" and rep["params_equal"] and rep["buckets"] >= 3
    assert rep["exchange_bytes"] > 0 and rep["launch_order"] == list(range(rep["buckets"]))
    assert all(s["grad_rel_err"] == 0.0 for s in rep["steps"])


def test_bench_one_rank_over_rccl_reports_the_exchange():
    """`bench.py --force-exchange`: the same forced world_size-1 RCCL exchange inside the bench step, with the `exchange` block
    (allreduce_ms_alone, step_ms_without_exchange) in the line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-exchange", "--config", "opt-125m", "--batch", "4",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing", "--ref-batch", "0"], capture_output=True,
                       text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    ex = line["exchange"]
    assert line["n_gpus"] == 1 and ex["rccl_ranks"] == 1 and ex["backend"] == "nccl" and ex["forced"]
    assert ex["exchange_bytes_per_step_per_gpu"] > 0 and ex["allreduce_ms_alone"] >= 0 and ex["step_ms_without_exchange"] > 0


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")
def test_two_ranks_engine_over_rccl():
    rep = _run_worker("nccl")
    assert rep["world"] == 2 and rep["params_equal"]


def test_bench_two_ranks_on_one_gpu_gloo():
    """`bench.py --gpus 2` end to end on ONE GPU (MMGL_DIST_BACKEND=gloo, both ranks on device 0): the self-launch, the per-rank
    batches, the hook-launched exchange and the max-over-ranks timing of the scaling run, minus the xGMI wire."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MMGL_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "opt-125m", "--batch", "4", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing", "--ref-batch", "0"], capture_output=True, text=True, env=env,
                       timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2"
    assert line["exchange"]["exchange_bytes_per_step_per_gpu"] > 0


def test_bench_self_launch_refuses_a_mismatched_world(monkeypatch):
    """WORLD_SIZE from a launcher must agree with --gpus: a silent 1-rank run that prints n_gpus = 1 is exactly what the
    scaling measurement must never get."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
