"""The two tests that kept gemm4w.hip / gemm8h.hip correct while they were linked into libmmgl_hip.so (round 5).  They run against
a variant library built by tools/experiments/gemm_variants/build.sh with the dispatch patch applied (see README.md here); they are not
part of `pytest tests`."""
import torch


def test_half_tile_gemm_matches_torch_and_gemm8p():
    """gemm8h.hip (DESIGN 9.6c iv-b: the two wave groups of a CU on separate half-tiles of one tile column, a K-wrapping shared operand
    stream; measured 0.97-1.00x gemm8p and not adopted) stays correct: the probe's checks in a process that selects it, and -- the
    accumulation order of a half-tile is a ROTATION of gemm8p's K order, fp32 addition is not associative -- equality with gemm8p's
    output to bf16 rounding on a shape whose half-tiles start at different K offsets."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    env = dict(os.environ, MMGL_GEMM_8H="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "experiments", "gemm_variants", "gemm4w_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    code = ("import torch, sys; sys.path.insert(0, %r); from mmgl_amd import ops; g = torch.Generator(device='cuda').manual_seed(3); "
            "x = torch.randn(5120, 768, device='cuda', generator=g).bfloat16(); w = (torch.randn(512, 768, device='cuda', generator=g) * 0.04).bfloat16(); "
            "b = torch.randn(512, device='cuda', generator=g).bfloat16(); y = ops.gemm_nt(x, w, b); torch.save(y.cpu(), sys.argv[1])") % root
    import tempfile
    outs = []
    for flag in ("1", "0"):
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            subprocess.run([sys.executable, "-c", code, f.name], env=dict(os.environ, MMGL_GEMM_8H=flag), check=True, timeout=300)
            outs.append(torch.load(f.name))
    d = (outs[0].float() - outs[1].float()).abs()
    assert float(d.max()) <= 2 ** -7 * float(outs[1].float().abs().max()), float(d.max())      # one bf16 ulp of the largest output
    assert float((d > 0).float().mean()) < 0.2                                                 # most outputs round identically


def test_four_wave_gemm_matches_torch():
    """gemm4w.hip (the 512-register, four-wave kernel of DESIGN 9.6c iv-a; measured 0.87x gemm8p and not adopted) stays correct: the
    probe's checks -- ragged rows / columns, lm_head's 96-column last tile column, bias, scale -- in a process that selects it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    env = dict(os.environ, MMGL_GEMM_4W="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "experiments", "gemm_variants", "gemm4w_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


