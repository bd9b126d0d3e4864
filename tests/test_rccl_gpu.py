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


def test_bench_eight_ranks_launched_as_the_driver_does_gloo():
    """The N = 8 line of the scaling run, on the ONE GPU a test box has: the driver's own command line (`python -m
    torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...`) with
    MMGL_DIST_BACKEND=gloo, all eight ranks on device 0 -- eight different seeded batches, eight engines cutting the same >= 4
    buckets (layout check at construction), hook-launched all-reduces in index order, barrier + max-over-ranks timing, ONE JSON line
    from rank 0 as the LAST line of stdout (reference run_generation.py:265-266, 283, 317-319)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MMGL_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "opt-125m", "--batch", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-kernel-timing", "--no-protocol", "--no-batch-sweep", "--ref-batch", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1800)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    json_lines = [l for l in lines if l.startswith('{"metric"')]
    assert len(json_lines) == 1, f"exactly one JSON line, from rank 0 (got {len(json_lines)})"
    assert lines[-1] == json_lines[0], "the JSON line is the last line of stdout"
    line = json.loads(json_lines[0])
    assert line["n_gpus"] == 8 and line["config"]["parallelism"] == "dp8" and line["config"]["global_batch"] == 16
    assert line["scaling"] == "weak" and line["value"] > 0 and line["samples_per_sec_per_gpu"] == pytest.approx(line["value"] / 8, rel=1e-3)
    ex = line["exchange"]
    assert ex["rccl_ranks"] == 8 and ex["buckets"] >= 4 and ex["exchange_bytes_per_step_per_gpu"] > 0
    assert ex["wire_bytes_per_step_per_gpu"] == int(2 * 7 / 8 * ex["exchange_bytes_per_step_per_gpu"])
    assert ex["first_bucket_fraction_of_gradient"] <= 0.25


def test_bench_self_launch_refuses_a_mismatched_world(monkeypatch):
    """WORLD_SIZE from a launcher must agree with --gpus: a silent 1-rank run that prints n_gpus = 1 is exactly what the
    scaling measurement must never get."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_c_abi_comm_one_rank_over_rccl():
    """include/mmgl_hip.h's gradient-exchange entry points (mmgl_comm_unique_id / _init / mmgl_allreduce_sum / mmgl_allgather /
    mmgl_broadcast / mmgl_comm_destroy: the reference's NCCL process group, run_generation.py:283, 317-319, 608-616) driven through
    ctypes on the one GPU of the box: a 1-rank RCCL communicator, collectives on torch's current stream.  With one rank a sum, a
    gather and a broadcast are the identity: values must come back bit-identical, for every dtype the ABI admits."""
    import ctypes
    from mmgl_amd import _lib
    L = _lib.lib()
    torch.cuda.set_device(0)
    uid = (ctypes.c_char * 128)()
    _lib.check(L.mmgl_comm_unique_id(uid), "unique_id")
    comm = ctypes.c_void_p()
    _lib.check(L.mmgl_comm_init(0, 1, uid, ctypes.byref(comm)), "init")
    assert comm.value
    st = _lib.stream_ptr()
    try:
        for dt, code in ((torch.float32, _lib.F32), (torch.bfloat16, _lib.BF16), (torch.int64, 2)):
            x = (torch.randn(100003, device="cuda") * 50).to(dt)
            want = x.clone()
            _lib.check(L.mmgl_allreduce_sum(comm, _lib.ptr(x), x.numel(), code, st), "allreduce")
            out = torch.empty_like(x)
            _lib.check(L.mmgl_allgather(comm, _lib.ptr(x), _lib.ptr(out), x.numel(), code, st), "allgather")
            _lib.check(L.mmgl_broadcast(comm, _lib.ptr(x), x.numel(), code, 0, st), "broadcast")
            torch.cuda.synchronize()
            assert torch.equal(x, want) and torch.equal(out, want), dt
        with pytest.raises(ValueError):
            _lib.check(L.mmgl_allreduce_sum(comm, _lib.ptr(x), x.numel(), 7, st), "bad dtype")
    finally:
        _lib.check(L.mmgl_comm_destroy(comm), "destroy")


_COMM_WORKER = r"""
import ctypes, os, sys, time, torch
sys.path.insert(0, sys.argv[1])
from mmgl_amd import _lib
rank, world, path = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
torch.cuda.set_device(rank)
L = _lib.lib()
uid = (ctypes.c_char * 128)()
if rank == 0:
    _lib.check(L.mmgl_comm_unique_id(uid), "uid")
    open(path + ".tmp", "wb").write(bytes(uid)); os.rename(path + ".tmp", path)
else:
    while not os.path.exists(path): time.sleep(0.05)
    ctypes.memmove(uid, open(path, "rb").read(), 128)
comm = ctypes.c_void_p()
_lib.check(L.mmgl_comm_init(rank, world, uid, ctypes.byref(comm)), "init")
st = _lib.stream_ptr()
x = torch.full((4096,), float(rank + 1), device="cuda", dtype=torch.bfloat16)
_lib.check(L.mmgl_allreduce_sum(comm, _lib.ptr(x), x.numel(), _lib.BF16, st), "allreduce")
g = torch.empty(world * 8, device="cuda", dtype=torch.int64)
mine = torch.arange(8, device="cuda") + 100 * rank
_lib.check(L.mmgl_allgather(comm, _lib.ptr(mine), _lib.ptr(g), 8, 2, st), "allgather")
b = torch.full((16,), float(rank), device="cuda")
_lib.check(L.mmgl_broadcast(comm, _lib.ptr(b), 16, _lib.F32, 1, st), "broadcast")
torch.cuda.synchronize()
assert float(x[0]) == world * (world + 1) / 2 and (x == x[0]).all()
assert g.tolist() == [i + 100 * r for r in range(world) for i in range(8)]
assert (b == 1.0).all()
_lib.check(L.mmgl_comm_destroy(comm), "destroy")
print("ok", rank)
"""


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")
def test_c_abi_comm_two_ranks_over_rccl(tmp_path):
    """The same entry points with two ranks on two GPUs (unique id handed over through a file): sums, gathers, broadcast."""
    script = tmp_path / "w.py"
    script.write_text(_COMM_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", str(tmp_path / "uid")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0 and "ok" in out, err[-2000:]
