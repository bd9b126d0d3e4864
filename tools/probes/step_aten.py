"""Which python lines launch the step's remaining ATen kernels (adds, copies, casts): torch.profiler with stacks over one
train step of the bench configuration.   python tools/probes/step_aten.py [batch]"""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from mmgl_amd.distributed import DataParallelEngine  # noqa: E402
from mmgl_amd.model import CrossAttentionModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = bench.CONFIGS["opt-1.3b"]
lm_cfg, txt_cfg, vis_cfg = bench.hf_configs(cfg)
margs = bench.make_args(cfg)
torch.manual_seed(1234)
device = torch.device("cuda", 0)
with torch.device("cpu"):
    model = CrossAttentionModel(margs, tokenizer=None, lm_config=lm_cfg, text_config=txt_cfg, visual_config=vis_cfg)
model = model.to(torch.bfloat16).to(device).train()
engine = DataParallelEngine(model, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.01)
batch, _ = bench.synthetic_batch(B, cfg, seed=1234, device=device)


T = batch["input_ids"].shape[1]
lin = cfg["lin"]


def step():
    out = model(**batch, logits_slice=slice(lin, T - 1))
    lg = out.logits.detach()
    torch.nn.functional.cross_entropy(lg.reshape(-1, lg.size(-1)).float(), batch["labels"][..., lin + 1:].reshape(-1), ignore_index=1).item()
    out.loss.backward()
    engine.finish_backward()
    engine.step()
    engine.zero_grad()


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::empty_like", "aten::view", "aten::reshape", "aten::as_strided", "aten::empty_strided", "aten::detach", "aten::alias", "aten::slice", "aten::select", "aten::transpose", "aten::t", "aten::unsqueeze", "aten::expand", "aten::_unsafe_view", "aten::result_type", "aten::item", "aten::_local_scalar_dense", "aten::to", "aten::contiguous", "aten::clone", "aten::resolve_conj", "aten::resolve_neg", "aten::lift_fresh", "aten::permute", "aten::unbind", "aten::squeeze", "aten::narrow", "aten::flatten", "aten::numel", "aten::size", "aten::stride", "aten::is_nonzero", "aten::new_empty", "aten::new_zeros", "aten::zeros", "aten::ones", "aten::full", "aten::arange", "aten::cumsum_", "aten::type_as", "aten::view_as", "aten::chunk", "aten::split", "aten::split_with_sizes", "aten::unflatten", "aten::pad", "aten::constant_pad_nd"):
        st = [f for f in (ev.stack or []) if "mmgl_amd" in f or "bench" in f or "transformers" in f]
        key = (ev.name, st[0] if st else "?")
        agg[key][0] += 1
        agg[key][1] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
for (name, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t:9.1f} us x{n:4d}  {name:18s} {where}")
