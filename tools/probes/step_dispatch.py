"""Which python lines issue the step's leftover ATen copies / adds / casts: a TorchDispatchMode over one train step that groups
them by (op, shape, innermost mmgl_amd frame).  Ops issued by the autograd engine itself (gradient accumulation) have no frame.
    python tools/probes/step_dispatch.py [config] [batch]"""
import os
import sys
import traceback
from collections import defaultdict

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from mmgl_amd.distributed import DataParallelEngine  # noqa: E402
from mmgl_amd import model as M  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "opt-1.3b"
cfg = bench.CONFIGS[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else cfg.get("batch", 16)
lm_cfg, txt_cfg, vis_cfg = bench.hf_configs(cfg)
margs = bench.make_args(cfg)
torch.manual_seed(1234)
device = torch.device("cuda", 0)
cls = M.SelfAttentionModel if cfg.get("kind") == "lora" else M.CrossAttentionModel
if cfg["kind"] == "llama":
    torch.set_default_dtype(torch.bfloat16)
    with torch.device(device):
        model = cls(margs, tokenizer=None, lm_config=lm_cfg, text_config=txt_cfg, visual_config=vis_cfg)
    torch.set_default_dtype(torch.float32)
else:
    with torch.device("cpu"):
        model = cls(margs, tokenizer=None, lm_config=lm_cfg, text_config=txt_cfg, visual_config=vis_cfg)
with torch.no_grad():
    for n_, p in model.named_parameters():
        if n_.endswith("gating1") or n_.endswith("gating2"):
            p.fill_(0.5)
model = model.to(torch.bfloat16).to(device).train()
engine = DataParallelEngine(model, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.01)
batch, _ = bench.synthetic_batch(B, cfg, seed=1234, device=device)
T = batch["input_ids"].shape[1]
lin = cfg["lin"]


def step():
    out = model(**batch, logits_slice=slice(lin, T - 1)) if cls is M.CrossAttentionModel else model(**batch)
    out.loss.backward()
    engine.finish_backward()
    engine.step()
    engine.zero_grad()


WATCH = ("copy_", "add", "add_", "_to_copy", "clone", "cat", "mul", "fill_", "zero_", "index_select", "sum", "contiguous")
NO_KERNEL = ("view", "empty", "empty_like", "slice", "detach", "as_strided", "alias", "_unsafe_view", "t", "transpose", "unsqueeze", "expand",
             "permute", "select", "reshape", "squeeze", "empty_strided", "new_empty", "unbind", "split", "narrow", "unflatten", "_reshape_alias",
             "view_as", "new_empty_strided", "lift_fresh", "split_with_sizes", "chunk")
ALL_OPS = os.environ.get("STEP_DISPATCH_ALL", "0") == "1"      # every ATen op with a >= 64k-element output, not just the WATCH list
agg = defaultdict(lambda: [0, 0])


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        nm = func.__name__.split(".")[0]
        if nm in WATCH or (ALL_OPS and nm not in NO_KERNEL):
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            if t is not None and t.is_cuda and t.numel() >= 1 << 16:
                fr = [f for f in traceback.extract_stack() if "mmgl_amd" in f.filename]
                where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}" if fr else "(autograd engine)"
                k = (nm, tuple(t.shape), str(t.dtype).replace("torch.", ""), where)
                agg[k][0] += 1
                agg[k][1] += t.numel() * t.element_size()
        return out


for _ in range(2):
    step()
torch.cuda.synchronize()
with Log():
    step()
torch.cuda.synchronize()
tot = sum(v[1] for v in agg.values())
print(f"{name} B={B}: {sum(v[0] for v in agg.values())} watched ATen calls on >= 64k-element tensors, {tot / 1e9:.2f} GB of outputs")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{v[1] / 1e6:9.1f} MB x{v[0]:4d}  {k[0]:12s} {str(k[1]):24s} {k[2]:9s} {k[3]}")
