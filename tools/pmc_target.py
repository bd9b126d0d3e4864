"""Target of tools/pmc_xattn.sh: a few launches of the cross-attention core at config-3 shape + a calibration copy."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_xattn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if sys.argv[2:3] == ["llama"]:                 # config 5's shape
    bench_xattn.run(B, H=32, T=2176, S=128, D=128, iters=10)
else:
    bench_xattn.run(B, iters=10)
src = torch.empty(128 << 20, dtype=torch.bfloat16, device="cuda").normal_()        # 256 MiB
dst = torch.empty_like(src)
for _ in range(5):
    dst.copy_(src)
torch.cuda.synchronize()
print("calibration copy: 268435456 bytes read + 268435456 bytes written per launch")
