"""Data-parallel gradient exchange + fused optimizer for the neighbor-fusion path (one process per GPU).

Replaces `torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu], find_unused_parameters=False)` +
`torch.optim.AdamW` of the reference (language_modelling/run_generation.py:317-333, 484-494) with an
xGMI-shaped design:

  * every trainable parameter is a view into ONE flat parameter buffer and its .grad a view into ONE flat gradient
    buffer, laid out in expected grad-ready order (last cross-attention layer first), so autograd accumulates
    straight into communication-ready memory and the optimizer is a single fused HIP kernel over the flat buffers
    (mmgl_adamw_step) instead of ~100 small per-tensor launches;
  * the flat gradient is cut into few LARGE contiguous buckets (default: an eighth of the gradient volume, 1..256 MiB -- 54 MB at
    config 3; xGMI is point-to-point, 7 links x ~153 GB/s per GPU, so RCCL's ring/direct algorithms want big messages, not DDP's
    25 MiB NVSwitch-era buckets, but the FIRST bucket must close early in backward for the exchange to hide behind the rest);
    a bucket's all-reduce is launched from a post-accumulate-grad hook the moment its last gradient lands, so the
    exchange overlaps the rest of backward on RCCL's own stream.  Only the LAST-ready bucket has nothing left to hide
    behind, so the tail of the flat buffer (the lowest gated layer and the neighbor projections) is cut off as its own
    small bucket (`tail_mb`, default min(32 MiB, bucket): ~1 ms on one xGMI link at 2 GPUs);
  * gradients are exchanged ONCE PER OPTIMIZER STEP (set `sync=False` on the non-final micro-batches): the reference
    all-reduces on every micro-batch because it never uses no_sync() (run_generation.py:484-485); the sum is the same
    up to summation order, the wire traffic is grad_accumulation_steps x smaller;
  * the 1/world_size averaging is folded into the optimizer kernel's grad_scale: no extra pass over the gradients.

`torch.distributed` (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) carries the collectives.
"""
import re
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def _grad_ready_order(named_params):
    """neighbor_layers.N descending (backward reaches the last cross-attention layer first), everything else
    (neighbor projections, poolers, embeddings: fed by ALL cross-attention layers, so ready last) afterwards."""
    pat = re.compile(r"neighbor_layers\.(\d+)\.")
    def key(item):
        idx, (name, _) = item
        m = pat.search(name)
        return (0, -int(m.group(1)), -idx) if m else (1, 0, -idx)
    return [np for _, np in sorted(enumerate(named_params), key=key)]


class DataParallelEngine:
    def __init__(self, model: torch.nn.Module, lr: float = 1e-3, betas=(0.9, 0.95), eps: float = 1e-8,
                 weight_decay: float = 0.01, bucket_mb: Optional[float] = None, tail_mb: Optional[float] = None, process_group=None, master_weights: Optional[bool] = None,
                 fused: Optional[bool] = None, broadcast: bool = True, optimizer: str = "adamw", force_exchange: Optional[bool] = None):
        import os
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # `exchange`: do the hooks launch collectives?  Always with more than one rank; with ONE rank only when asked
        # (force_exchange / MMGL_DDP_FORCE_EXCHANGE=1 and an initialised process group): a world_size-1 RCCL group runs the whole
        # device-side path -- hook-launched async all-reduces on RCCL's stream, work.wait(), the dynamic GEMM tile schedule -- on
        # the one GPU a test box has (tests/test_rccl_gpu.py, bench.py --force-exchange)
        if force_exchange is None:
            force_exchange = os.environ.get("MMGL_DDP_FORCE_EXCHANGE", "0") == "1"
        self.exchange = self.world > 1 or (bool(force_exchange) and dist.is_initialized())
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.sync = True
        self.optimizer_kind = optimizer

        named = _grad_ready_order([(n, p) for n, p in model.named_parameters() if p.requires_grad])
        if not named:
            raise ValueError("DataParallelEngine: the model has no trainable parameters")
        seen, uniq = set(), []
        for n, p in named:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append((n, p))
        self.names = [n for n, _ in uniq]
        self.params = [p for _, p in uniq]
        p0 = self.params[0]
        self.device, self.dtype = p0.device, p0.dtype
        if any(p.dtype != self.dtype or p.device != self.device for p in self.params):
            raise ValueError("DataParallelEngine: trainable parameters must share one dtype and device")
        self.fused = self.device.type == "cuda" if fused is None else fused
        dyn = os.environ.get("MMGL_GEMM_DYNAMIC", "1")       # "0": static always; "1": dynamic when collectives share the GPU; "2": always
        if self.device.type == "cuda" and ((self.exchange and dyn != "0") or dyn == "2"):
            from . import ops
            ops.gemm_dynamic_schedule(True, self.device)          # the GEMMs of the backward pass share the CUs with the all-reduces
        if master_weights is None:
            master_weights = self.dtype != torch.float32

        # ---- flat buffers (each tensor starts on a 128-element boundary: 256/512-byte aligned views)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 127) // 128 * 128
        self.numel = off
        self.flat_param = torch.zeros(off, dtype=self.dtype, device=self.device)
        self.flat_grad = torch.zeros(off, dtype=self.dtype, device=self.device)
        for p, o in zip(self.params, self.offsets):
            self.flat_param[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o:o + p.numel()].view(p.shape)
            p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.master = self.flat_param.float() if master_weights else None
        self.torch_optimizer = None
        if optimizer == "adafactor":          # the reference trains T5 with Adafactor (run_generation.py:321-324)
            from transformers.optimization import Adafactor
            self.torch_optimizer = Adafactor(self.params, scale_parameter=False, relative_step=False, warmup_init=False, lr=lr)
        elif optimizer != "adamw":
            raise ValueError(f"unknown optimizer {optimizer!r}")

        # ---- buckets = contiguous ranges of the flat gradient, closed in grad-ready order
        # bucket size from the gradient volume (bucket_mb=None): an eighth of the flat gradient, at most 256 MiB, at least 1 MiB.  A fixed
        # 256 MiB (rounds 1-5) held config 3's first all-reduce back until ~60 % of the backward pass (433 MB of bf16 gradients = 1.7
        # buckets); an eighth closes the first bucket inside the LAST gated layer's backward (a gated layer is ~23 % of the trainable
        # set at every config) and keeps >= 8 collectives in flight behind the rest of backward, each still tens of MB at OPT-1.3B
        # (xGMI is point-to-point: ring steps want MBs, not DDP's 25 MiB/8-way split -- 54 MB buckets are 6.8 MB per ring step at 8 GPUs)
        total_mb = self.numel * self.flat_grad.element_size() / (1 << 20)
        if bucket_mb is None:
            bucket_mb = min(256.0, max(1.0, total_mb / 8))
        if tail_mb is None:
            tail_mb = min(32.0, bucket_mb)
        self.bucket_mb, self.tail_mb = bucket_mb, tail_mb
        cap = max(1, int(bucket_mb * (1 << 20)) // self.flat_grad.element_size())
        self.buckets: List[Dict] = []
        start, members = 0, []
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            members.append(i)
            end = self.offsets[i + 1] if i + 1 < len(self.params) else self.numel
            if end - start >= cap or i + 1 == len(self.params):
                self.buckets.append(dict(start=start, end=end, members=members, pending=len(members), work=None))
                start, members = end, []
        # the last-ready bucket is the exposed one: keep at most tail_mb of it, hand the rest to a bucket of its own
        tail = int(tail_mb * (1 << 20)) // self.flat_grad.element_size()
        last = self.buckets[-1] if self.buckets else None
        if last and tail > 0 and last["end"] - last["start"] > tail and len(last["members"]) > 1:
            ms = last["members"]
            cut = len(ms) - 1                                  # first member of the tail bucket
            while cut > 1 and last["end"] - self.offsets[ms[cut - 1]] <= tail:
                cut -= 1
            head, rest = ms[:cut], ms[cut:]
            mid = self.offsets[rest[0]]
            self.buckets[-1:] = [dict(start=last["start"], end=mid, members=head, pending=len(head), work=None),
                                 dict(start=mid, end=last["end"], members=rest, pending=len(rest), work=None)]
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            b["idx"] = bi
            for i in b["members"]:
                self._bucket_of[i] = b
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))
        self.exchange_bytes = 0
        # Collectives must be issued in the same order on every rank (RCCL matches them by issue order, not by buffer).  The order
        # is FIXED: bucket k is issued only after buckets 0..k-1 (a bucket that fills early waits for its predecessors; the flat
        # buffer is laid out in grad-ready order, so in practice nothing waits), and the bucket LAYOUT -- the one thing that could
        # still differ between ranks, if they built different models -- is compared across ranks here, before the first
        # collective on gradient memory is ever issued.  A diagnostic after the fact (rounds 2-3) could only fire once RCCL had
        # already matched mismatched all-reduces.
        self.launch_order: List[int] = []
        self.last_launch_order: List[int] = []      # of the last finished exchange (tests, diagnostics)
        self._next_launch = 0
        if self.exchange:
            self._check_bucket_layout()
        if broadcast and self.exchange:              # DDP's constructor broadcast of parameters AND buffers (run_generation.py:319)
            dist.broadcast(self.flat_param, src=0, group=self.pg)    # (after the layout check: a size mismatch must raise, not abort)
            mine = {id(p) for p in self.params}
            seen_t = set()
            for t in list(model.parameters()) + list(model.buffers()):
                if id(t) in mine or id(t) in seen_t or t.numel() == 0:
                    continue
                seen_t.add(id(t))
                dist.broadcast(t.data, src=0, group=self.pg)      # frozen LM / encoders: one-off, ~3 GB at OPT-1.3B
            if self.master is not None:
                self.master.copy_(self.flat_param)

    # ---------------------------------------------------------------------------------- gradient exchange
    def _make_hook(self, i):
        def hook(param):
            want = self.flat_grad[self.offsets[i]:self.offsets[i] + param.numel()]
            if param.grad is None or param.grad.data_ptr() != want.data_ptr():
                # autograd replaced the view (first accumulation into an undefined grad): fold it back into the flat buffer
                if param.grad is not None:
                    want.add_(param.grad.reshape(-1).to(want.dtype))
                param.grad = want.view(param.shape)
            b = self._bucket_of[i]
            b["pending"] -= 1
            if b["pending"] == 0 and self.sync and self.exchange:
                self._launch_ready()
        return hook

    def _launch_ready(self):
        """Issue every filled bucket whose predecessors have all been issued: index order, on every rank."""
        while self._next_launch < len(self.buckets) and self.buckets[self._next_launch]["pending"] == 0:
            self._launch(self.buckets[self._next_launch])

    def _launch(self, b):
        assert b["idx"] == self._next_launch, "bucket all-reduces are issued in index order"
        b["work"] = dist.all_reduce(self.flat_grad[b["start"]:b["end"]], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self.exchange_bytes += (b["end"] - b["start"]) * self.flat_grad.element_size()
        self.launch_order.append(b["idx"])
        self._next_launch += 1

    def _check_bucket_layout(self):
        """Every rank must cut the flat gradient into the same buckets (same model, same trainable set, same bucket_mb): one small
        all-gather of the bucket boundaries at construction time."""
        mine = torch.tensor([self.numel, len(self.buckets)] + [b["end"] for b in self.buckets], dtype=torch.int64, device=self.device)
        sizes = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
        dist.all_gather(sizes, torch.tensor([mine.numel()], dtype=torch.int64, device=self.device), group=self.pg)
        if any(int(t) != mine.numel() for t in sizes):
            raise RuntimeError(f"DataParallelEngine: ranks built different numbers of gradient buckets: {[int(t) - 2 for t in sizes]}")
        both = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(both, mine, group=self.pg)
        layouts = [t.tolist() for t in both]
        if any(l != layouts[0] for l in layouts[1:]):
            raise RuntimeError(f"DataParallelEngine: the gradient buckets differ across ranks (different trainable sets?): {layouts}")

    def finish_backward(self):
        """Call after loss.backward(): waits for the bucket all-reduces (if this was a sync step) and re-arms the hooks.
        A parameter that received no gradient leaves its bucket open: launched here (find_unused_parameters=False is a
        contract of the reference, so this is the rare path)."""
        if self.sync and self.exchange:
            for b in self.buckets[self._next_launch:]:
                self._launch(b)
            for b in self.buckets:
                b["work"].wait()
        self.last_launch_order, self.launch_order, self._next_launch = self.launch_order, [], 0
        for b in self.buckets:
            b["pending"], b["work"] = len(b["members"]), None

    def zero_grad(self):
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):       # optimizer.zero_grad(set_to_none) callers must not detach the views
            if p.grad is None or p.grad.data_ptr() != self.flat_grad[o:o + p.numel()].data_ptr():
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)

    # ---------------------------------------------------------------------------------- optimizer
    def step(self, lr: Optional[float] = None):
        """AdamW (torch.optim.AdamW formula) over the flat buffers; gradients are averaged over ranks via grad_scale."""
        lr = self.lr if lr is None else lr
        self.step_count += 1
        scale = 1.0 / self.world
        b1, b2 = self.betas
        if self.torch_optimizer is not None:
            if self.world > 1:
                self.flat_grad.mul_(scale)
            for g in self.torch_optimizer.param_groups:
                g["lr"] = lr
            self.torch_optimizer.step()
            return
        if self.fused:
            from . import ops
            ops.adamw_step_(self.flat_param, self.master, self.flat_grad, self.exp_avg, self.exp_avg_sq, lr, b1, b2, self.eps,
                            self.weight_decay, self.step_count, scale)
            return
        # reference formula in plain torch (CPU tests of the exchange logic only)
        g = self.flat_grad.float() * scale
        p = self.master if self.master is not None else self.flat_param
        p.mul_(1 - lr * self.weight_decay)
        self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        denom = (self.exp_avg_sq.sqrt() / bc2 ** 0.5).add_(self.eps)
        p.addcdiv_(self.exp_avg, denom, value=-lr / bc1)
        if self.master is not None:
            self.flat_param.copy_(self.master)

    def grad_norm(self) -> torch.Tensor:
        return self.flat_grad.float().norm() / self.world

    # ---------------------------------------------------------------------------------- checkpointing
    def _model_param_names(self):
        """Names in `model.parameters()` order -- the index space of the reference's torch.optim.AdamW(model.parameters())
        (run_generation.py:328): tied parameters appear once, frozen ones are counted."""
        return [n for n, _ in self.model.named_parameters()]

    def state_dict(self):
        """torch.optim.AdamW.state_dict() layout over ALL of model.parameters() in model order (reference ckpt 'optimizer',
        :411): `state` holds the entries of the trainable parameters at their model index (frozen parameters never get state
        in torch either), `param_groups[0]['params']` lists every index, so `torch.optim.AdamW(model.parameters())
        .load_state_dict()` on the reference side accepts it.  `param_names` (index -> name) is an extra key torch ignores."""
        names = self._model_param_names()
        index = {n: i for i, n in enumerate(names)}
        state = {}
        for n, p, o in zip(self.names, self.params, self.offsets):
            k = p.numel()
            state[index[n]] = dict(step=torch.tensor(float(self.step_count)), exp_avg=self.exp_avg[o:o + k].view(p.shape).clone(),
                                   exp_avg_sq=self.exp_avg_sq[o:o + k].view(p.shape).clone())
        group = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, amsgrad=False, maximize=False,
                     foreach=None, capturable=False, differentiable=False, fused=None, params=list(range(len(names))))
        return dict(state=state, param_groups=[group], param_names=names)

    def load_state_dict(self, sd):
        """Accepts this engine's checkpoints and the reference's (torch AdamW over model.parameters()): optimizer state is
        matched BY PARAMETER NAME -- through the checkpoint's own `param_names` when present, else through this model's
        named_parameters() order, which is the order the reference's optimizer indexed.  A checkpoint that carries state but
        matches none of the trainable parameters is an error, never a silent cold start."""
        names = sd.get("param_names") or self._model_param_names()
        index = {n: i for i, n in enumerate(names)}
        state = sd.get("state", {})
        matched, missing = 0, []
        for n, p, o in zip(self.names, self.params, self.offsets):
            st = state.get(index[n]) if n in index else None
            if st is None:
                missing.append(n)
                continue
            k = p.numel()
            if st["exp_avg"].numel() != k:
                raise ValueError(f"optimizer state of {n!r} has {st['exp_avg'].numel()} elements, the parameter has {k}")
            self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(st["step"])
            matched += 1
        if state and not matched:
            raise ValueError("optimizer checkpoint carries state for none of the trainable parameters (index space mismatch): "
                             f"checkpoint indices {sorted(state)[:5]}..., trainable names {self.names[:3]}...")
        if missing and matched:
            import warnings
            warnings.warn(f"optimizer checkpoint has no state for {len(missing)} trainable parameter(s), e.g. {missing[:3]}: "
                          "their Adam moments start from zero")
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        if self.master is not None:
            self.master.copy_(self.flat_param.float())

    def sync_master_from_params(self):
        """After load_state_dict on the model: refresh the fp32 master copy."""
        if self.master is not None:
            self.master.copy_(self.flat_param.float())
