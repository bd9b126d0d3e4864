"""How many host threads should the CPU oracle baseline use?  (one-off probe, results in DESIGN.md)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
print("cpu_count", os.cpu_count())
x = torch.randn(640, 2048); w = torch.randn(8192, 2048)
for nt in (16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    F.linear(x, w)
    t = time.time()
    for _ in range(5): F.linear(x, w)
    dt = (time.time() - t) / 5
    print(nt, "threads: linear 640x2048x8192", round(dt * 1e3, 2), "ms", round(2 * 640 * 2048 * 8192 / dt / 1e9, 1), "GFLOP/s")
