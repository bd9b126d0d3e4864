"""CPU tests of the N>1 path: world_size-2 gloo processes exercising DataParallelEngine (flat buckets, exchange once
per optimizer step, averaged AdamW) against single-process training on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class Toy(nn.Module):
    """Parameter names follow the real model so the grad-ready ordering logic is exercised."""
    def __init__(self):
        super().__init__()
        self.neighbor_layers = nn.ModuleList([nn.Linear(12, 12) for _ in range(3)])
        self.text_embeddings = nn.Linear(6, 12)
        self.frozen = nn.Linear(12, 12)
        for p in self.frozen.parameters():
            p.requires_grad = False

    def forward(self, x):
        h = self.text_embeddings(x)
        for l in self.neighbor_layers:
            h = torch.tanh(l(self.frozen(h))) + h
        return h.pow(2).mean()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reference_run(data, accum, steps, lr):
    torch.manual_seed(0)
    m = Toy()
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=lr, betas=(0.9, 0.95), weight_decay=0.01, eps=1e-8)
    it = 0
    for s in range(steps):
        opt.zero_grad()
        for a in range(accum):
            # mean over ranks of per-rank micro-batch losses, each / accum
            loss = sum(m(data[r][it]) for r in range(2)) / 2 / accum
            loss.backward()
            it += 1
        opt.step()
    return torch.cat([p.detach().reshape(-1) for p in m.parameters() if p.requires_grad])


def _worker(rank, world, port, data, accum, steps, lr, bucket_mb, out, tail_mb=32):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmgl_amd.distributed import DataParallelEngine
    torch.manual_seed(0 if rank == 0 else 123)         # rank 1 starts from different weights: the constructor must broadcast
    m = Toy()
    eng = DataParallelEngine(m, lr=lr, betas=(0.9, 0.95), weight_decay=0.01, bucket_mb=bucket_mb, tail_mb=tail_mb, fused=False)
    assert eng.names[0].startswith("neighbor_layers.2") and eng.names[-1].startswith("text_embeddings")
    it = 0
    for s in range(steps):
        eng.zero_grad()
        for a in range(accum):
            eng.sync = (a == accum - 1)
            (m(data[rank][it]) / accum).backward()
            eng.finish_backward()
            it += 1
        eng.step()
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters() if p.requires_grad])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save(dict(params=flat, other=gathered[1], exchange_bytes=eng.exchange_bytes, numel=eng.numel, nbuckets=len(eng.buckets)), out)
    dist.destroy_process_group()


@pytest.mark.parametrize("accum,bucket_mb", [(1, 256), (3, 256), (2, 0)])
def test_two_rank_engine_matches_single_process(tmp_path, accum, bucket_mb):
    steps, lr = 3, 1e-2
    g = torch.Generator().manual_seed(5)
    data = [[torch.randn(4, 6, generator=g) for _ in range(steps * accum)] for _ in range(2)]
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, nprocs=2, args=(2, _free_port(), data, accum, steps, lr, bucket_mb, out), join=True)
    res = torch.load(out)
    ref = _reference_run(data, accum, steps, lr)
    assert torch.equal(res["params"], res["other"]), "ranks diverged"
    assert torch.allclose(res["params"], ref, rtol=1e-5, atol=1e-6), (res["params"] - ref).abs().max()
    # exactly one exchange of the flat gradient per OPTIMIZER step, independent of the accumulation depth
    assert res["exchange_bytes"] == steps * res["numel"] * 4
    if bucket_mb == 0:
        assert res["nbuckets"] > 1


def test_two_rank_engine_with_a_tail_bucket(tmp_path):
    """One big bucket whose last-ready part is cut off as a small tail bucket (the exposed exchange): same result."""
    steps, lr, accum = 3, 1e-2, 2
    g = torch.Generator().manual_seed(6)
    data = [[torch.randn(4, 6, generator=g) for _ in range(steps * accum)] for _ in range(2)]
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, nprocs=2, args=(2, _free_port(), data, accum, steps, lr, 256, out, 1e-4), join=True)
    res = torch.load(out)
    ref = _reference_run(data, accum, steps, lr)
    assert res["nbuckets"] == 2
    assert torch.equal(res["params"], res["other"]), "ranks diverged"
    assert torch.allclose(res["params"], ref, rtol=1e-5, atol=1e-6), (res["params"] - ref).abs().max()
    assert res["exchange_bytes"] == steps * res["numel"] * 4


def test_tail_bucket_layout():
    """Buckets tile the flat gradient without gaps, in order; the last one holds at most tail_mb (or one parameter)."""
    from mmgl_amd.distributed import DataParallelEngine
    torch.manual_seed(0)
    m = torch.nn.Sequential(*[torch.nn.Linear(64, 64) for _ in range(12)])
    eng = DataParallelEngine(m, fused=False, bucket_mb=0.05, tail_mb=0.02)       # 16.6 KB per layer
    bs = eng.buckets
    assert bs[0]["start"] == 0 and bs[-1]["end"] == eng.numel and all(a["end"] == b["start"] for a, b in zip(bs, bs[1:]))
    assert [i for b in bs for i in b["members"]] == list(range(len(eng.params)))
    assert len(bs) >= 2 and (bs[-1]["end"] - bs[-1]["start"]) * 4 <= 0.02 * (1 << 20)
    for b in bs:
        assert b["start"] == eng.offsets[b["members"][0]]


def test_default_bucket_size_follows_the_gradient_volume():
    """bucket_mb=None: an eighth of the flat gradient (1..256 MiB).  A gated-layer-shaped model -- 4 equal layers named
    neighbor_layers.N, a projection ready last -- gets >= 4 buckets, the first of which closes INSIDE the last layer's backward (it
    holds only parameters of neighbor_layers.3), so the first all-reduce starts a quarter of the way into backward at the latest."""
    from mmgl_amd.distributed import DataParallelEngine

    class Layer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1, self.fc2 = torch.nn.Linear(256, 1024), torch.nn.Linear(1024, 256)
            self.q, self.o = torch.nn.Linear(256, 256), torch.nn.Linear(256, 256)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = torch.nn.Linear(256, 1024)
            self.neighbor_layers = torch.nn.ModuleList([Layer() for _ in range(4)])

    eng = DataParallelEngine(Net(), fused=False)
    total_mb = eng.numel * 4 / (1 << 20)
    assert abs(eng.bucket_mb - max(1.0, total_mb / 8)) < 1e-9 and eng.tail_mb == min(32.0, eng.bucket_mb)
    assert len(eng.buckets) >= 4
    first = [eng.names[i] for i in eng.buckets[0]["members"]]
    assert all(n.startswith("neighbor_layers.3.") for n in first), first
    layer3 = sum(p.numel() for n, p in zip(eng.names, eng.params) if n.startswith("neighbor_layers.3."))
    assert eng.buckets[0]["end"] - eng.buckets[0]["start"] < layer3
    # explicit sizes are honoured as before
    assert len(DataParallelEngine(Net(), fused=False, bucket_mb=256, tail_mb=0).buckets) == 1


def test_engine_single_process_state_dict_roundtrip():
    from mmgl_amd.distributed import DataParallelEngine
    torch.manual_seed(0)
    m = Toy()
    eng = DataParallelEngine(m, lr=1e-2, fused=False)
    for _ in range(2):
        eng.zero_grad()
        m(torch.randn(4, 6)).backward()
        eng.finish_backward()
        eng.step()
    sd = eng.state_dict()
    assert len(sd["state"]) == len(eng.params) and sd["param_groups"][0]["betas"] == (0.9, 0.95)
    m2 = Toy()
    m2.load_state_dict(m.state_dict())
    eng2 = DataParallelEngine(m2, lr=1e-2, fused=False)
    eng2.load_state_dict(sd)
    x = torch.randn(4, 6)
    for e, mm in ((eng, m), (eng2, m2)):
        e.zero_grad()
        mm(x).backward()
        e.finish_backward()
        e.step()
    for a, b in zip(m.parameters(), m2.parameters()):
        assert torch.allclose(a, b, atol=1e-7)
    # every trainable parameter is a view into the flat buffers
    for p, o in zip(eng.params, eng.offsets):
        assert p.data_ptr() == eng.flat_param[o:].data_ptr() and p.grad.data_ptr() == eng.flat_grad[o:].data_ptr()


def test_optimizer_state_interchanges_with_torch_adamw_over_all_parameters(tmp_path):
    """The reference builds torch.optim.AdamW(model.parameters()) over ALL parameters (run_generation.py:328) and stores its
    state_dict in the checkpoint (:411).  The engine's optimizer state must round-trip with that index space BY NAME in both
    directions, through torch.save / torch.load(weights_only=True)."""
    from mmgl_amd.distributed import DataParallelEngine
    torch.manual_seed(0)
    x = [torch.randn(5, 6) for _ in range(4)]
    # --- "reference side": AdamW over all parameters (frozen ones included), two steps
    m_ref = Toy()
    opt = torch.optim.AdamW(m_ref.parameters(), lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01, eps=1e-8)
    for t in range(2):
        opt.zero_grad()
        m_ref(x[t]).backward()
        opt.step()
    ck = tmp_path / "ref.pt"
    torch.save({"optimizer": opt.state_dict()}, ck)
    # --- engine resumes from the reference's optimizer state
    m = Toy()
    m.load_state_dict(m_ref.state_dict())
    eng = DataParallelEngine(m, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01, fused=False, master_weights=False)
    eng.load_state_dict(torch.load(ck, weights_only=True)["optimizer"])
    assert eng.step_count == 2
    for t in range(2, 4):                               # two more steps on both sides must stay identical
        opt.zero_grad(); m_ref(x[t]).backward(); opt.step()
        m(x[t]).backward(); eng.finish_backward(); eng.step(); eng.zero_grad()
    for (n, a), (_, b) in zip(m.named_parameters(), m_ref.named_parameters()):
        assert torch.allclose(a, b, atol=1e-6), n
    # --- and back: torch AdamW over all parameters accepts the engine's state_dict
    ck2 = tmp_path / "eng.pt"
    torch.save({"optimizer": eng.state_dict()}, ck2)
    sd = torch.load(ck2, weights_only=True)["optimizer"]
    opt2 = torch.optim.AdamW(m.parameters(), lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01, eps=1e-8)
    opt2.load_state_dict(sd)
    names = [n for n, _ in m.named_parameters()]
    ref_state = opt.state_dict()["state"]
    for i, st in opt2.state_dict()["state"].items():
        assert torch.allclose(st["exp_avg"], ref_state[i]["exp_avg"], atol=1e-6), names[i]
        assert int(st["step"]) == 4
    # --- a checkpoint in a foreign index space (state but no matching entry) is an error, not a silent cold start
    bad = {"state": {10_000: dict(step=torch.tensor(1.0), exp_avg=torch.zeros(1), exp_avg_sq=torch.zeros(1))},
           "param_groups": sd["param_groups"], "param_names": None}
    with pytest.raises(ValueError):
        eng.load_state_dict(bad)


def test_lr_schedule_follows_the_reference_step_order():
    """optimizer.step() runs at the current lr, then scheduler.step() (run_generation.py:486-494): the warm-up's first optimizer
    step uses lr 0 and step n uses base * (n - 1) / warmup."""
    from mmgl_amd.language_modelling.run_generation import WarmupStepLR
    s = WarmupStepLR(base_lr=1.0, warmup=4, step_size=3, gamma=0.5)
    used = []
    for _ in range(9):
        used.append(s.get_last_lr()[0])     # what train_loop hands to engine.step()
        s.step()
    assert used == [0.0, 0.25, 0.5, 0.75, 1.0, 1.0, 1.0, 0.5, 0.5]


def test_bf16_gradient_sum_over_eight_ranks_error_bound():
    """The engine all-reduces gradients in the model's dtype (bf16 under --bf16, as the reference's DDP does: run_generation.py:304-319).
    What an 8-way bf16 ring sum costs against an fp32 sum of the same bf16 gradients: every hop rounds a partial sum to bf16 (2^-9
    relative), so the element-wise error stays below 8 * 2^-9 of the largest partial sum and the norm-wise error of the averaged
    gradient near 2^-9 -- the same order as the bf16 rounding each rank's own gradient already carries."""
    g = torch.Generator().manual_seed(7)
    n, world = 1 << 18, 8
    common = torch.randn(n, generator=g)                                           # the signal the ranks agree on
    grads = [(common + 0.5 * torch.randn(n, generator=g)).bfloat16() for _ in range(world)]
    exact = torch.stack([x.float() for x in grads]).sum(0) / world
    acc = grads[0].clone()
    for x in grads[1:]:                                                            # ring reduce: partial sums travel in bf16
        acc = (acc.float() + x.float()).bfloat16()
    got = acc.float() / world
    rel = float((got - exact).norm() / exact.norm())
    worst = float((got - exact).abs().max() / exact.abs().max())
    assert rel <= 2.0 ** -8 and worst <= world * 2.0 ** -9, (rel, worst)


def _order_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmgl_amd.distributed import DataParallelEngine
    torch.manual_seed(0)
    m = Toy()
    eng = DataParallelEngine(m, lr=1e-2, bucket_mb=0, tail_mb=0, fused=False, force_exchange=True)
    assert eng.exchange and len(eng.buckets) == len(eng.params) >= 4
    # fire the hooks in REVERSE grad-ready order (what a rank with a different autograd schedule would do): the engine must
    # still issue bucket 0 first, then 1, ... -- the all-reduces of a filled bucket wait for its predecessors
    for p in eng.params:
        p.grad.fill_(1.0)
    seen = []
    for i in reversed(range(len(eng.params))):
        eng._make_hook(i)(eng.params[i])
        seen.append(list(eng.launch_order))
    assert seen[0] == [] and seen[-2] == [], seen              # nothing may go out before bucket 0 is full
    assert seen[-1] == list(range(len(eng.buckets)))
    eng.finish_backward()
    assert eng.last_launch_order == list(range(len(eng.buckets)))
    assert torch.equal(eng.flat_grad[:eng.params[0].numel()], torch.full((eng.params[0].numel(),), float(world)))
    if rank == 0:
        torch.save(dict(ok=True), out)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_bucket_all_reduces_go_out_in_index_order_whatever_the_hook_order(tmp_path, world):
    """the order check ran after the collectives it was meant to protect.  Now the order is fixed by construction;
    world = 1 with force_exchange is the configuration tests/test_rccl_gpu.py runs over RCCL."""
    out = str(tmp_path / "ok.pt")
    mp.spawn(_order_worker, nprocs=world, args=(world, _free_port(), out), join=True)
    assert torch.load(out)["ok"]


def _layout_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmgl_amd.distributed import DataParallelEngine
    torch.manual_seed(0)
    m = Toy()
    if rank == 1:
        m.neighbor_layers[1].bias.requires_grad = False       # a different trainable set on one rank
    try:
        DataParallelEngine(m, lr=1e-2, bucket_mb=256, fused=False)
        raised = False
    except RuntimeError as e:
        raised = "differ across ranks" in str(e) or "different numbers" in str(e)
    torch.save(dict(raised=raised), f"{out}.{rank}")
    dist.destroy_process_group()


def test_mismatched_bucket_layout_raises_at_construction(tmp_path):
    out = str(tmp_path / "layout")
    mp.spawn(_layout_worker, nprocs=2, args=(2, _free_port(), out), join=True)
    assert torch.load(f"{out}.0")["raised"] and torch.load(f"{out}.1")["raised"]
