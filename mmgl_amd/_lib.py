"""ctypes binding of libmmgl_hip.so (C ABI in include/mmgl_hip.h).

The library must exist: there is NO CPU / eager fallback anywhere in this package.  A missing or
unloadable .so raises at first use, loudly.
"""
import ctypes
import os

import torch  # imported first on purpose: the .so must bind to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMGL_LIB_PATH") or os.path.join(_HERE, "libmmgl_hip.so")     # override: timing experiments with ablated builds

ABI_VERSION = 104         # = mmgl_version() of the library this binding was written against (csrc/lib.hip)
F32, BF16 = 0, 1
ACT_NONE, ACT_RELU = 0, 1
_ERR_INVALID, _ERR_UNSUPPORTED, _ERR_HIP = 1, 2, 3

c_void_p, c_int, c_float, c_size_t, c_u64, c_i64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t,
                                                    ctypes.c_uint64, ctypes.c_int64)
P, I, F, Z, U, L = c_void_p, c_int, c_float, c_size_t, c_u64, c_i64

# name -> (restype, argtypes); every symbol include/mmgl_hip.h declares
SIGNATURES = {
    "mmgl_last_error": (ctypes.c_char_p, []),
    "mmgl_version": (I, []),
    "mmgl_xattn_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "mmgl_xattn_bwd_workspace": (Z, [I, I, I, I, I]),
    "mmgl_xattn_bwd": (I, [P, P, P, P, P, P, P, P, P, P, Z, I, I, I, I, I, I, P]),
    "mmgl_selfattn_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "mmgl_selfattn_bwd_workspace": (Z, [I, I, I]),
    "mmgl_selfattn_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, Z, I, I, I, I, I, I, I, P]),
    "mmgl_selfattn_prefix_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "mmgl_selfattn_prefix_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, Z, I, I, I, I, I, I, I, I, I, I, P]),
    "mmgl_attn_general_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, U, I, P]),
    "mmgl_attn_general_bwd_workspace": (Z, [I, I, I]),
    "mmgl_attn_general_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, Z, I, I, I, I, I, I, F, U, I, P]),
    "mmgl_attn_dropout_mask": (I, [P, I, I, I, I, F, U, P]),
    "mmgl_layernorm_fwd": (I, [P, P, P, P, P, P, I, I, F, I, P]),
    "mmgl_norm_bwd_workspace": (Z, [I, I]),
    "mmgl_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, Z, I, I, I, P]),
    "mmgl_rmsnorm_fwd": (I, [P, P, P, P, I, I, F, I, P]),
    "mmgl_rmsnorm_bwd": (I, [P, P, P, P, P, P, P, Z, I, I, I, P]),
    "mmgl_add_rmsnorm_fwd": (I, [P, P, P, P, P, P, I, I, F, I, P]),
    "mmgl_add_rmsnorm_bwd": (I, [P, P, P, P, P, P, P, P, Z, I, I, I, P]),
    "mmgl_gated_residual_fwd": (I, [P, P, P, P, Z, F, U, I, P]),
    "mmgl_gated_residual_bwd_workspace": (Z, [Z]),
    "mmgl_gated_residual_bwd": (I, [P, P, P, P, P, P, Z, Z, F, U, I, P]),
    "mmgl_linear_fwd": (I, [P, P, P, P, I, I, I, I, F, I, P]),
    "mmgl_linear_dgrad_workspace": (Z, [I, I, I, I, I]),
    "mmgl_linear_dgrad": (I, [P, P, P, P, P, Z, I, I, I, I, F, I, P]),
    "mmgl_linear_wgrad_workspace": (Z, [I, I, I, I]),
    "mmgl_linear_wgrad": (I, [P, P, P, P, P, P, Z, I, I, I, I, F, I, I, P]),
    "mmgl_linear_bwd_workspace": (Z, [I, I, I, I, I]),
    "mmgl_linear_bwd": (I, [P, P, P, P, P, P, P, P, Z, I, I, I, I, F, I, I, I, P]),
    "mmgl_transpose": (I, [P, P, I, I, I, P]),
    "mmgl_lora_linear_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, F, I, P]),
    "mmgl_lora_linear_bwd_workspace": (Z, [I, I, I, I, I]),
    "mmgl_lora_linear_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, Z, I, I, I, I, F, I, I, P]),
    "mmgl_neighbor_interleave_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "mmgl_neighbor_interleave_bwd": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "mmgl_cross_entropy_fwd": (I, [P, P, P, P, P, P, I, I, L, I, P]),
    "mmgl_cross_entropy_bwd": (I, [P, P, P, P, P, P, I, I, L, I, P]),
    "mmgl_position_ids": (I, [P, P, I, I, P]),
    "mmgl_adamw_step": (I, [P, P, P, P, P, Z, F, F, F, F, F, I, F, I, P]),
    "mmgl_encattn_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "mmgl_add_layernorm_fwd": (I, [P, P, P, P, P, P, P, P, I, I, F, F, U, I, P]),
    "mmgl_add_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, Z, I, I, F, U, I, P]),
    "mmgl_activation_fwd": (I, [P, P, Z, I, I, P]),
    "mmgl_scale": (I, [P, P, P, Z, I, P]),
    "mmgl_gemm_nt_fast": (I, [I, I, I, I, I, I, I]),
    "mmgl_gemm_nt_workspace": (Z, [I, I, I, I, I, I, I]),
    "mmgl_gemm_nt_relu_bits_bytes": (Z, [I, I, I, I, I, I, I]),
    "mmgl_gemm_nt_relu_bits": (I, [P, I, P, I, P, P, I, P, I, I, I, F, I, P]),
    "mmgl_gemm_nt_masked": (I, [P, I, P, I, P, P, I, I, I, I, F, I, P]),
    "mmgl_gemm_nt": (I, [P, I, P, I, P, P, P, P, I, I, I, I, I, F, P, Z, I, P]),
    "mmgl_relu_bwd": (I, [P, P, P, Z, I, P]),
    "mmgl_gemm_set_tile_counter": (I, [P]),
    "mmgl_gemm_get_tile_counter": (P, []),
    "mmgl_rope_inplace": (I, [P, P, Z, I, I, I, I, I, I, I, P]),
    "mmgl_rope": (I, [P, P, P, Z, I, I, I, I, I, I, I, I, P]),
    "mmgl_swiglu_fwd": (I, [P, P, Z, I, I, P]),
    "mmgl_swiglu_bwd": (I, [P, P, P, Z, I, I, P]),
    "mmgl_comm_unique_id": (I, [P]),
    "mmgl_comm_init": (I, [I, I, P, ctypes.POINTER(ctypes.c_void_p)]),
    "mmgl_allreduce_sum": (I, [P, P, Z, I, P]),
    "mmgl_allgather": (I, [P, P, P, Z, I, P]),
    "mmgl_broadcast": (I, [P, P, Z, I, I, P]),
    "mmgl_comm_destroy": (I, [P]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m mmgl_amd._build` "
                "(or __graft_entry__.build()). mmgl_amd has no CPU fallback.")
        h = ctypes.CDLL(LIB_PATH)
        h.mmgl_version.restype = ctypes.c_int
        if h.mmgl_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports ABI version {h.mmgl_version()}, this binding expects {ABI_VERSION}: a stale build -- "
                               "run `python -m mmgl_amd._build --force`")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc: int, what: str = ""):
    if rc == 0:
        return
    msg = lib().mmgl_last_error().decode("utf-8", "replace")
    if rc in (_ERR_INVALID, _ERR_UNSUPPORTED):
        raise ValueError(msg or what)
    raise RuntimeError(msg or what)


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise ValueError(f"mmgl_amd kernels take float32 or bfloat16 activations, got {t.dtype}")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def ptr_off(t, nbytes):
    """Device pointer `nbytes` past the start of t (a column slice of a fused buffer)."""
    return ctypes.c_void_p(t.data_ptr() + int(nbytes))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """torch's current stream of the current device as a hipStream_t.  Through torch._C directly: torch.cuda.current_stream() builds a
    Stream object through four layers of Python (9 us per call, 3 ms per step of ~3200 C-ABI calls at the launch-bound batch sizes)."""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mmgl_amd ops run on the GPU only (tensor is on %s); there is no CPU path" % t.device)


class KernelTimer:
    """Optional per-entry-point timing with HIP events on torch's current stream (= the stream the kernels are launched
    on).  Off by default (zero overhead); bench.py turns it on to price each C-ABI call against its roofline."""
    enabled = False
    only = None           # optional set of entry-point names: time just these (bench.py's timed region: the roofline kernel)
    records = []          # (name, start_event, end_event, work dict)
    by_tag = bool(os.environ.get("MMGL_KERNEL_TAGS"))     # split the table by the calls' `tag` (GEMM shapes) as well

    @classmethod
    def reset(cls):
        cls.records = []

    @classmethod
    def summary(cls):
        """name -> dict(calls, ms_total, ms_avg, bytes, flops) after a device synchronize."""
        torch.cuda.synchronize()
        out = {}
        for name, s, e, work in cls.records:
            if cls.by_tag and "tag" in work:
                name = f"{name}[{work['tag']}]"
            d = out.setdefault(name, dict(calls=0, ms_total=0.0, bytes=0.0, flops=0.0))
            d["calls"] += 1
            d["ms_total"] += s.elapsed_time(e)
            d["bytes"] += work.get("bytes", 0.0)
            d["flops"] += work.get("flops", 0.0)
        for d in out.values():
            d["ms_avg"] = d["ms_total"] / d["calls"]
        return out


def call(name: str, work, *args):
    """Invoke C-ABI entry point `name`; raises ValueError / RuntimeError on a non-zero return code."""
    fn = getattr(lib(), name)
    if KernelTimer.enabled and (KernelTimer.only is None or name in KernelTimer.only):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        KernelTimer.records.append((name, s, e, work() if callable(work) else (work or {})))
    else:
        rc = fn(*args)
    check(rc, name)
