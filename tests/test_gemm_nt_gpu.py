"""mmgl_gemm_nt (persistent ping-pong MFMA kernel, csrc/gemm8p.hip) against fp32 torch on the frozen path's real shapes:
every epilogue, ragged M / N edges, strided operands, the FFN pair with the ReLU backward in fc2's dgrad epilogue and the
lm_head dgrad whose contraction length (vocab) is zero-padded.  Tolerance: bf16 rounding of inputs/outputs, fp32 accumulate."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mk(M, N, K, seed, bias=True, resid=False, zmask=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16() if bias else None
    r = torch.randn(M, N, device="cuda", generator=g).bfloat16() if resid else None
    z = torch.randn(M, N, device="cuda", generator=g).bfloat16() if zmask else None
    return x, W, b, r, z


def _ref(x, W, b, r, z, act, scale):
    v = x.float() @ W.float().t()
    if b is not None:
        v = v + b.float()
    v = v * scale
    v = {0: lambda t: t, 1: torch.relu, 2: F.gelu, 3: lambda t: t * torch.sigmoid(1.702 * t), 4: lambda t: F.gelu(t, approximate="tanh")}[act](v)
    if z is not None:
        v = torch.where(z.float() > 0, v, torch.zeros_like(v))
    if r is not None:
        v = v + r.float()
    return v


CASES = [
    # M, N, K, act, bias, resid, zmask, scale            (>= 160 tiles of 256x256 -> fast path)
    (40960, 2048, 2048, 1, True, False, False, 1.0),     # config-3 fc-style shapes at the bench batch
    (40960, 6144, 2048, 0, True, False, False, 1.0),     # fused QKV
    (10240, 8192, 2048, 1, True, False, False, 1.0),     # fc1 + ReLU
    (10240, 2048, 8192, 0, False, False, True, 0.5),     # fc2 dgrad-like: zmask epilogue, scale
    (40000, 2000, 768, 2, True, False, False, 1.0),      # ragged M and N tiles, GELU(erf): RoBERTa FFN
    (20003, 3072, 768, 3, True, True, False, 1.0),       # quick-GELU + residual: CLIP FFN
    (5000, 50272, 2048, 0, False, False, False, 1.0),    # lm_head forward (N = vocab, last tile 96 columns)
    (4099, 4096, 11008, 4, True, True, True, 1.0),       # config-5 dims, tanh-GELU, everything at once
    (2560, 2048, 8192, 1, True, False, False, 1.0),      # the reference's batch of 4 (80 tiles): K-split work items + finish kernel
    (2600, 2048, 6144, 3, True, True, False, 0.5),       # uneven K splits (48 steps over 3), ragged M, residual, scale
    (2560, 2064, 4096, 0, False, False, True, 1.0),      # K split with a ragged last column tile and the zmask applied by the finish kernel
    (2560, 2064, 3072, 0, False, False, True, 1.0),      # same at K = 3072: 90 tiles -> the 128x128 kernel since round 6 (gemm8p_plan)
    (2560, 8192, 2048, 1, True, False, False, 1.0),      # 320 tiles on 256 CUs: 2048 rows = one round of the persistent kernel + 512 tail rows on the few-tile kernel
    (2560, 8192, 2048, 0, True, True, True, 0.5),        # the same row split with zmask + residual (both offset to the tail rows) and a scale
    (2400, 8192, 2048, 1, True, True, True, 1.0),        # the row split with a ragged tail (2048 + 352 rows), every epilogue stage
    (4700, 8192, 2048, 3, True, False, False, 1.0),      # two whole rounds (4096 rows) + a 604-row tail
    (2570, 8200, 2048, 0, True, True, True, 0.5),        # the same with ragged last tile row / column (363 tiles), zmask + residual in the finish kernel
    (2560, 8192, 4096, 3, False, False, False, 1.0),     # the row split at K = 4096, quick-GELU
    (17408, 4096, 4096, 0, True, True, False, 1.0),      # config 5 (8 x 2176 rows): 1088 tiles = 4.25 rounds -> 4 whole rounds + 64 tiles as 4 K splits each
    (1024, 512, 256, 1, True, False, False, 1.0),        # few tiles: the 128x128 kernel (csrc/gemm_mid.hip)
    (300, 96, 64, 2, True, True, True, 2.0),             # tiny, one K step, ragged rows and columns, every epilogue stage
    (2560, 2048, 2048, 0, True, True, False, 1.0),       # out_proj at the reference's batch of 4: 320 tiles of 128x128, two workgroups per CU
    (2563, 2056, 2048, 4, True, False, True, 0.5),       # the same, ragged last tile row / column
    (333, 136, 192, 1, False, True, True, 1.0),          # three K steps, no bias
    (2560, 2048, 1984, 3, True, False, False, 1.0),      # K a multiple of 64 only (31 steps): not the persistent kernel's
]


@pytest.mark.parametrize("M,N,K,act,bias,resid,zmask,scale", CASES)
def test_gemm_nt_vs_fp32(M, N, K, act, bias, resid, zmask, scale):
    from mmgl_amd import ops
    x, W, b, r, z = _mk(M, N, K, seed=M + N + K, bias=bias, resid=resid, zmask=zmask)
    buf = torch.full((M + 4, N), float("nan"), device="cuda", dtype=torch.bfloat16)      # guard rows past M must stay untouched
    y = ops.gemm_nt(x, W, b, r, z, act=act, out_scale=scale, out=buf[:M])
    want = _ref(x, W, b, r, z, act, scale)
    err = (y.float() - want).abs().max().item()
    tol = 2e-2 * want.abs().max().item() + 1e-2
    assert torch.isfinite(y.float()).all()
    assert err <= tol, f"max err {err} > {tol}"
    assert torch.isnan(buf[M:].float()).all(), "rows past M were written"


def test_gemm_nt_fast_path_is_selected():
    from mmgl_amd import _lib
    L = _lib.lib()
    assert L.mmgl_gemm_nt_fast(40960, 2048, 2048, 2048, 2048, 2048, _lib.BF16) == 1
    assert L.mmgl_gemm_nt_fast(40960, 2048, 2048, 2048, 2048, 2048, _lib.F32) == 0
    assert L.mmgl_gemm_nt_fast(1024, 512, 256, 256, 256, 512, _lib.BF16) == 2          # too few 256x256 tiles: the 128x128 kernel
    assert L.mmgl_gemm_nt_fast(2560, 2048, 2048, 2112, 2048, 2176, _lib.BF16) == 2       # ... which takes strided operands too
    assert L.mmgl_gemm_nt_fast(40960, 2048, 1984, 1984, 1984, 2048, _lib.BF16) == 2      # K % 128 != 0, K % 64 == 0
    assert L.mmgl_gemm_nt_fast(40960, 2048, 2000, 2000, 2000, 2048, _lib.BF16) == 0      # K % 64: composed path


@pytest.mark.parametrize("M,N,K", [(40960, 2048, 1024), (2563, 2048, 1024), (77, 72, 64)])
def test_gemm_nt_strided_operands(M, N, K):
    """x and W as column slices of wider buffers (ldx, ldw > K), y into a column slice (ldy > N): the persistent kernel and the
    128x128 kernel."""
    from mmgl_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    xb = torch.randn(M, K + 256, device="cuda", generator=g).bfloat16()
    Wb = (torch.randn(N, K + 128, device="cuda", generator=g) * K ** -0.5).bfloat16()
    yb = torch.zeros(M, N + 64, device="cuda", dtype=torch.bfloat16)
    x, W = xb[:, 128:128 + K], Wb[:, 64:64 + K]
    ops.gemm_nt(x, W, out=yb[:, 32:32 + N])
    want = x.float() @ W.float().t()
    assert (yb[:, 32:32 + N].float() - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 1e-2
    assert float(yb[:, :32].abs().max()) == 0 and float(yb[:, 32 + N:].abs().max()) == 0


def test_gemm_nt_padded_contraction_on_the_128_kernel():
    """K past x's row length against zero columns of W (mmgl_gemm_nt's contract, the lm_head dgrad's trick) on the 128x128 kernel:
    a row's tail reads the head of the next row (times zero), the LAST row's tail must not read past the tensor (NaN planted there)."""
    from mmgl_amd import ops
    M, N, K0, K = 300, 136, 200, 256
    g = torch.Generator(device="cuda").manual_seed(11)
    buf = torch.full((M + 1, K0), float("nan"), device="cuda", dtype=torch.bfloat16)
    buf[:M] = torch.randn(M, K0, device="cuda", generator=g).bfloat16()
    x = buf[:M]
    W = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16)
    W[:, :K0] = (torch.randn(N, K0, device="cuda", generator=g) * K0 ** -0.5).bfloat16()
    y = ops.gemm_nt(x, W, K=K)
    want = x.float() @ W[:, :K0].float().t()
    assert bool(torch.isfinite(y).all())
    assert (y.float() - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 1e-2


@pytest.mark.parametrize("M,d,ffn", [(40960, 2048, 8192), (2560, 2048, 8192), (200, 64, 128)])
def test_frozen_ffn_pair_mask_dx(M, d, ffn):
    """fc1 + ReLU (bwd_premasked) -> fc2 (mask_dx): forward and dx against fp32 torch autograd (reference :352-355 frozen)."""
    from mmgl_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(M, d, device="cuda", generator=g).bfloat16().requires_grad_()
    W1 = (torch.randn(ffn, d, device="cuda", generator=g) * d ** -0.5).bfloat16()
    b1 = (torch.randn(ffn, device="cuda", generator=g) * 0.1).bfloat16()
    W2 = (torch.randn(d, ffn, device="cuda", generator=g) * ffn ** -0.5).bfloat16()
    b2 = (torch.randn(d, device="cuda", generator=g) * 0.1).bfloat16()
    w = torch.randn(M, d, device="cuda", generator=g).bfloat16()
    h = ops.frozen_linear(x, W1, b1, relu=True, bwd_premasked=True)
    if M >= 2560:                    # whole tiles of the persistent kernel: the ReLU mask travels as bits, h is not kept for backward
        assert getattr(h, "_mmgl_relu_bits", None) is not None
    y = ops.frozen_linear(h, W2, b2, mask_dx=True)
    if M >= 2560:
        assert y.grad_fn.mask_bits is not None
        kept = y.grad_fn.saved_tensors[2]
        if M == 2560:                # 320 tiles on 256 CUs: 2048 rows as one round of the persistent kernel (mask bits), 512 tail rows on the few-tile
            assert y.grad_fn.mask_bits[2] == 2048 and tuple(kept.shape) == (512, ffn)      # kernel with their activation rows as the mask
        else:
            assert kept is None
    (y.float() * w.float()).sum().backward()
    xr = x.detach().float().requires_grad_()
    hr = torch.relu(F.linear(xr, W1.float(), b1.float()))
    # the kernel masks with the bf16-rounded h it stored: reproduce that mask in the reference (sign flips at pre-activation ~ 0)
    hr_m = hr * (h.detach().float() > 0)
    yr = F.linear(hr_m, W2.float(), b2.float())
    (yr * w.float()).sum().backward()
    assert (y.float() - yr).abs().max().item() <= 3e-2 * yr.abs().max().item() + 1e-2
    assert (x.grad.float() - xr.grad).abs().max().item() <= 3e-2 * xr.grad.abs().max().item() + 1e-2


@pytest.mark.parametrize("M,N,K,pitch", [(40960, 8192, 2048, 8320), (5000, 8192, 2048, 8192), (2570, 8208, 2048, 8208)])
def test_relu_mask_bits_round_trip(M, N, K, pitch):
    """mmgl_gemm_nt_relu_bits writes (y > 0) as lane-private bits, mmgl_gemm_nt_masked applies them to another product of the same
    [M, N]: equal to masking with the stored activation itself (ragged last tiles, padded row pitch)."""
    from mmgl_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W1 = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b1 = (torch.randn(N, device="cuda", generator=g) * 0.1).bfloat16()
    dy = torch.randn(M, 2048, device="cuda", generator=g).bfloat16()
    W2t = (torch.randn(N, 2048, device="cuda", generator=g) * 2048 ** -0.5).bfloat16()      # = W2^T: [ffn, d_out]
    assert ops.relu_bits_bytes(M, N, K, K, K, pitch, torch.bfloat16) == ((M + 255) // 256) * ((N + 255) // 256) * 8192
    h = torch.empty(M, pitch, device="cuda", dtype=torch.bfloat16)[:, :N]
    _, bits = ops.gemm_nt_relu_bits(x, W1, b1, h)
    want_h = ops.gemm_nt(x, W1, b1, act=1)
    assert torch.equal(h, want_h)
    dh = torch.full((M, pitch), float("nan"), device="cuda", dtype=torch.bfloat16)[:, :N]
    ops.gemm_nt_masked(dy, W2t, bits, dh)
    want = ops.gemm_nt(dy, W2t, zmask=want_h.contiguous())
    assert torch.equal(dh, want)
    assert ops.relu_bits_bytes(1000, 512, 256, 256, 256, 512, torch.bfloat16) == 0            # too few tiles: no bits
    assert ops.relu_bits_bytes(M, N, K, K, K, pitch, torch.float32) == 0


def test_frozen_lm_head_padded_dgrad():
    """dx = dlogits @ W with the contraction over the vocabulary (50272, not a multiple of 128): W^T is zero-padded to 50304
    columns and dlogits is read with its own row stride (reference :826 lm_head, frozen / tied)."""
    from mmgl_amd import ops
    M, d, V = 2560, 2048, 50272
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(M, d, device="cuda", generator=g).bfloat16().requires_grad_()
    W = (torch.randn(V, d, device="cuda", generator=g) * d ** -0.5).bfloat16()
    dl = (torch.randn(M, V, device="cuda", generator=g) * 0.01).bfloat16()
    y = ops.frozen_linear(x, W, None)
    y.backward(dl)
    want_y = x.detach().float() @ W.float().t()
    want_dx = dl.float() @ W.float()
    assert (y.float() - want_y).abs().max().item() <= 2e-2 * want_y.abs().max().item() + 1e-2
    assert (x.grad.float() - want_dx).abs().max().item() <= 2e-2 * want_dx.abs().max().item() + 1e-3


@pytest.mark.parametrize("M,d,V,chunk", [(40960, 2048, 50272, 8192), (2176 * 4, 4096, 32000, 4096), (100, 64, 128, 32)])
def test_lm_head_cross_entropy_fused_equals_unfused(M, d, V, chunk):
    """lm_head + token cross-entropy without the [rows, V] logits (reference :826-836) against the unfused pair (frozen_linear +
    mmgl_cross_entropy): loss to 1e-3, d hidden to 2e-2 (bf16), including ignored rows and an upstream gradient != 1."""
    from mmgl_amd import ops
    g = torch.Generator(device="cuda").manual_seed(V)
    h = (torch.randn(M, d, device="cuda", generator=g) * 0.5).bfloat16()
    W = (torch.randn(V, d, device="cuda", generator=g) * d ** -0.5).bfloat16()
    lab = torch.randint(0, V, (M,), device="cuda", generator=g)
    lab[::7] = -100
    h1 = h.clone().requires_grad_()
    loss1 = ops.lm_head_cross_entropy(h1, W, lab, chunk_rows=chunk)
    (loss1 * 0.37).backward()
    h2 = h.clone().requires_grad_()
    loss2 = ops.cross_entropy(ops.frozen_linear(h2, W, None), lab)
    (loss2 * 0.37).backward()
    assert abs(float(loss1) - float(loss2)) <= 1e-3 * abs(float(loss2))
    assert (h1.grad.float() - h2.grad.float()).abs().max().item() <= 2e-2 * h2.grad.float().abs().max().item()
    ref = torch.nn.functional.cross_entropy(h.float() @ W.float().t(), lab, ignore_index=-100) if M <= 10000 else None
    if ref is not None:
        assert abs(float(loss1) - float(ref)) <= 5e-3 * abs(float(ref))


def test_training_step_does_not_build_logits():
    """MPTForCausalLM in train mode with a frozen head: the default is the reference's contract (.logits [B, T, V]); with
    return_logits=False the step returns the loss without building them; `logits_slice` returns just those positions."""
    from helpers import mpt_args, tiny_opt_config
    from mmgl_amd.model.modelling_cross_attention import MPTConfig, MPTForCausalLM
    torch.manual_seed(0)
    lm = MPTForCausalLM(MPTConfig(mpt_args(), tiny_opt_config(dropout=0.0))).cuda()
    ids = torch.randint(3, 128, (2, 24), device="cuda")
    am = torch.ones_like(ids)
    ne = torch.randn(2, 6, 64, device="cuda")
    nv = torch.ones(2, 6, dtype=torch.bool, device="cuda")
    kw = dict(input_ids=ids, attention_mask=am, labels=ids, neighbor_embeds=ne, neighbor_attention_mask=nv)
    lm.train()
    assert lm(**kw).logits.shape == (2, 24, 128)
    o = lm(**kw, return_logits=False)
    assert o.logits is None and o.loss.requires_grad
    o.loss.backward()
    assert any(p.grad is not None and float(p.grad.abs().max()) > 0 for n, p in lm.named_parameters() if "neighbor_layers" in n)
    o2 = lm(**kw, logits_slice=slice(16, -1))
    lm.eval()
    with torch.no_grad():
        oe = lm(**kw)
    assert oe.logits.shape == (2, 24, 128) and o2.logits.shape == (2, 7, 128)
    assert abs(float(o.loss) - float(oe.loss)) < 1e-4 * abs(float(oe.loss))
    assert (o2.logits.float() - oe.logits[:, 16:-1].float()).abs().max() < 1e-4
    lm.train()
    o3 = lm(**kw, return_logits=True)
    assert o3.logits.shape == (2, 24, 128)


@pytest.mark.parametrize("M,N,K,act,bias,resid,zmask,scale", [CASES[i] for i in (0, 1, 3, 4, 6, 8, 11, 12)])
def test_dynamic_tile_schedule_is_bitwise_the_static_one(M, N, K, act, bias, resid, zmask, scale):
    """mmgl_gemm_set_tile_counter: tiles handed out through per-XCD atomic counters instead of `tile = i * grid + workgroup` --
    several rounds of tiles, ragged edges, K-split items, the hybrid plan.  Same tiles, same arithmetic per tile => bit-identical
    output; every launch leaves the counters at zero (the next launch depends on it)."""
    from mmgl_amd import ops
    x, W, b, r, z = _mk(M, N, K, seed=M + N + K, bias=bias, resid=resid, zmask=zmask)
    y0 = ops.gemm_nt(x, W, b, r, z, act=act, out_scale=scale)
    ctr = ops.gemm_dynamic_schedule(True)
    try:
        y1 = ops.gemm_nt(x, W, b, r, z, act=act, out_scale=scale)
        y2 = ops.gemm_nt(x, W, b, r, z, act=act, out_scale=scale)
        torch.cuda.synchronize()
        assert int(ctr.abs().sum()) == 0, ctr.tolist()
    finally:
        ops.gemm_dynamic_schedule(False)
    assert torch.equal(y0, y1) and torch.equal(y0, y2)


@pytest.mark.parametrize("M", [8192, 12288, 16384, 20480, 24576])
def test_dynamic_tile_schedule_one_to_three_rounds(M):
    """every workgroup claimed three items before computing anything, so an output of 1 / 1.5 / 2 / 2.5 rounds of tiles
    (256 / 384 / 512 / 640 tiles on 256 CUs) was computed by a fraction of the CUs: 3 tile times where the static schedule takes
    1 / 2 / 2 / 3.  Up-front claims are capped at ceil(items / workgroups) now: bit-identical output, counters left at zero, and the
    dynamic launch takes no longer than the static one (a gross check: 1.35x; the defect was 1.5x - 3x)."""
    from mmgl_amd import ops
    N = K = 2048
    x, W, b, _, _ = _mk(M, N, K, seed=M, bias=True)
    y0 = ops.gemm_nt(x, W, b, act=1)

    def timed():
        for _ in range(3):
            ops.gemm_nt(x, W, b, act=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = ops.gemm_nt(x, W, b, act=1)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10, y

    t_static, _ = timed()
    ctr = ops.gemm_dynamic_schedule(True)
    try:
        t_dyn, y1 = timed()
        assert int(ctr.abs().sum()) == 0, ctr.tolist()
    finally:
        ops.gemm_dynamic_schedule(False)
    assert torch.equal(y0, y1)
    print(f"M={M}: static {t_static * 1e3:.1f} us, dynamic {t_dyn * 1e3:.1f} us")
    assert t_dyn < 1.35 * t_static, (t_static, t_dyn)


def test_dynamic_tile_schedule_is_bound_to_one_stream():
    """include/mmgl_hip.h: while a tile counter is set, the device's counters belong to ONE stream (they are shared); a launch on a
    second stream runs on the static schedule instead (it never touches the counters), so side-stream work stays legal and correct;
    setting the counter again releases the binding.  (MMGL_GEMM_STRICT_STREAM=1 turns the side-stream launch into MMGL_ERR_INVALID.)"""
    from mmgl_amd import ops
    x, W, b, _, _ = _mk(8192, 2048, 2048, seed=5, bias=True)
    y0 = ops.gemm_nt(x, W, b)
    side = torch.cuda.Stream()
    ctr = ops.gemm_dynamic_schedule(True)
    try:
        y1 = ops.gemm_nt(x, W, b)                    # binds the counters to the current stream
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            y5 = ops.gemm_nt(x, W, b)                # another stream: static schedule, counters untouched
        torch.cuda.synchronize()
        assert int(ctr.abs().sum()) == 0, ctr.tolist()
        y2 = ops.gemm_nt(x, W, b)                    # the bound stream keeps working
        ops.gemm_dynamic_schedule(True)              # set again: binding released
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            y3 = ops.gemm_nt(x, W, b)
        torch.cuda.synchronize()
    finally:
        ops.gemm_dynamic_schedule(False)
    with torch.cuda.stream(side):                    # static schedule: any stream
        y4 = ops.gemm_nt(x, W, b)
    torch.cuda.synchronize()
    assert all(torch.equal(y0, y) for y in (y1, y2, y3, y4, y5))


def test_dynamic_tile_schedule_reaches_the_backward_thread():
    """The GEMMs that overlap the gradient all-reduce are the BACKWARD ones, and autograd launches those from its own device thread:
    the schedule set from the main thread must be what a launch from another thread sees (a thread-local setting left exactly those
    launches static)."""
    import threading
    from mmgl_amd import _lib, ops
    L = _lib.lib()
    seen = {}

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            seen["fwd"] = (threading.get_ident(), L.mmgl_gemm_get_tile_counter())
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            seen["bwd"] = (threading.get_ident(), L.mmgl_gemm_get_tile_counter())
            return g

    assert L.mmgl_gemm_get_tile_counter() is None
    ctr = ops.gemm_dynamic_schedule(True)
    try:
        x = torch.randn(8, device="cuda", requires_grad=True)
        Probe.apply(x).sum().backward()
        t = threading.Thread(target=lambda: seen.__setitem__("thread", (threading.get_ident(), L.mmgl_gemm_get_tile_counter())))
        t.start()
        t.join()
    finally:
        ops.gemm_dynamic_schedule(False)
    assert seen["bwd"][0] != seen["fwd"][0], "autograd ran backward on the calling thread: the test does not probe what it is meant to"
    assert seen["fwd"][1] == seen["bwd"][1] == seen["thread"][1] == ctr.data_ptr()
    assert L.mmgl_gemm_get_tile_counter() is None


def test_dynamic_tile_schedule_ffn_relu_bits():
    """The frozen FFN pair under the dynamic schedule: fc1 writes its ReLU mask as bits indexed by the tile's virtual id, fc2's dgrad
    reads them back -- whichever workgroup happened to take the tile."""
    from mmgl_amd import ops
    torch.manual_seed(0)
    x = torch.randn(10240, 2048, device="cuda").bfloat16().requires_grad_()
    w1 = (torch.randn(8192, 2048, device="cuda") * 0.02).bfloat16()
    b1 = torch.randn(8192, device="cuda").bfloat16()
    w2 = (torch.randn(2048, 8192, device="cuda") * 0.02).bfloat16()

    def run():
        y = ops.frozen_linear(ops.frozen_linear(x, w1, b1, relu=True, bwd_premasked=True), w2, None, mask_dx=True)
        (g,) = torch.autograd.grad(y.float().square().mean(), x)
        return y.detach(), g
    y0, g0 = run()
    ops.gemm_dynamic_schedule(True)
    try:
        y1, g1 = run()
    finally:
        ops.gemm_dynamic_schedule(False)
    assert torch.equal(y0, y1) and torch.equal(g0, g1)
