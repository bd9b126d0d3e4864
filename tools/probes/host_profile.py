#!/usr/bin/env python
"""Where does the HOST spend a training step?  cProfile over a few steps of bench.py's step at a small batch (the launch-bound regime:
opt-125m at B = 16 is ~1000 launches in ~16 ms).   python tools/probes/host_profile.py [config] [batch] [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "opt-125m"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    from mmgl_amd import _lib
    from mmgl_amd.distributed import DataParallelEngine
    from mmgl_amd.model import CrossAttentionModel
    cfg = bench.CONFIGS[name]
    lm, txt, vis = bench.hf_configs(cfg)
    torch.manual_seed(0)
    with torch.device("cpu"):
        model = CrossAttentionModel(bench.make_args(cfg), tokenizer=None, lm_config=lm, text_config=txt, visual_config=vis)
    model = model.bfloat16().cuda().train()
    engine = DataParallelEngine(model, lr=1e-4)
    batch, _ = bench.synthetic_batch(B, cfg, seed=1, device=torch.device("cuda"))
    T = batch["input_ids"].shape[1]
    sl = slice(cfg["lin"], T - 1)

    def step():
        out = model(**batch, logits_slice=sl)
        out.loss.backward()
        engine.finish_backward()
        engine.step()
        engine.zero_grad()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ncalls = [0]
    orig = _lib.call

    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{name} B={B}: host launch time {1e3 * t_host / steps:.2f} ms/step, wall {1e3 * t_all / steps:.2f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(60)


if __name__ == "__main__":
    main()
