#!/usr/bin/env python
"""Per-optimizer-step times of train_loop at a reference batch protocol (per_device x accum), fused or literal: the trainer's own
event-timed `examples_per_sec` with print_freq 1, several epochs over the same pinned micro-batches.
    python tools/probes/protocol_timing.py [config] [per_device] [accum] [fuse 0|1] [groups] [epochs]"""
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    a = sys.argv[1:]
    name = a[0] if a else "opt-1.3b"
    per_device, accum, fuse, groups, epochs = (int(a[i]) if len(a) > i else d for i, d in ((1, 4), (2, 4), (3, 1), (4, 8), (5, 3)))
    from mmgl_amd.distributed import DataParallelEngine
    from mmgl_amd.language_modelling.run_generation import WarmupStepLR, train_loop
    from mmgl_amd.model import CrossAttentionModel
    cfg = bench.CONFIGS[name]
    lm, txt, vis = bench.hf_configs(cfg)
    torch.manual_seed(0)
    with torch.device("cpu"):
        model = CrossAttentionModel(bench.make_args(cfg), tokenizer=None, lm_config=lm, text_config=txt, visual_config=vis)
    model = model.bfloat16().cuda().train()
    engine = DataParallelEngine(model, lr=1e-4)
    mbs = []
    for i in range(accum * groups):
        hb, _ = bench.synthetic_batch(per_device, cfg, seed=9000 + i, device=torch.device("cpu"))
        hb.pop("host_meta")
        mbs.append({k: v.pin_memory() for k, v in hb.items()})
    sched = WarmupStepLR(1e-4, 0, 1 << 30, 1.0)
    targs = SimpleNamespace(steps_per_epoch=accum * groups, grad_accumulation_steps=accum, decoder_only=True, max_input_length=cfg["lin"],
                            print_freq=1, per_device_train_batch_size=per_device, fuse_grad_accumulation=bool(fuse), fused_pass_tokens=49152)
    import contextlib
    import io
    for ep in range(epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            hist = train_loop(mbs, model, None, engine, ep, sched, targs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        per = [round(per_device * accum / (h["examples_per_sec"] / per_device) / per_device * 1e3 / accum * accum, 1) if h["examples_per_sec"] else None for h in hist]
        ms = [round(1e3 * per_device * accum / h["examples_per_sec"], 1) for h in hist]
        print(f"epoch {ep}: {per_device}x{accum} fuse={fuse}: {per_device * accum * groups / dt:.1f} samples/s wall; ms per optimizer step (event-timed): {ms}", flush=True)


if __name__ == "__main__":
    main()
