"""CPU tests of the drop-in boundary: libmmgl_hip.so loads and exports every symbol include/mmgl_hip.h declares (no
compute calls: there is no GPU here), argument validation returns the documented error codes, and the product path
fails loudly -- never silently falls back -- without a GPU / without the extension."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mmgl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmgl_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    from mmgl_amd import _lib
    names = _declared()
    assert len(names) >= 28
    h = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/mmgl_hip.h but not exported by libmmgl_hip.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in mmgl_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)
    assert _lib.lib().mmgl_version() == _lib.ABI_VERSION      # a stale .so (older ABI) is refused at load


def test_library_is_not_older_than_its_sources():
    """A failed compile leaves the previous libmmgl_hip.so in place and every later run silently measures old code (it happened:
    an inline-asm constraint the HOST pass rejected).  The library must be newer than every kernel source and header."""
    import glob
    from mmgl_amd import _lib
    srcs = glob.glob(os.path.join(ROOT, "mmgl_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "mmgl_amd", "csrc", "*.h")) \
        + [os.path.join(ROOT, "include", "mmgl_hip.h")]
    newest = max(srcs, key=os.path.getmtime)
    assert os.path.getmtime(_lib.LIB_PATH) >= os.path.getmtime(newest), \
        f"{_lib.LIB_PATH} is older than {newest}: run `python -m mmgl_amd._build` and read its output"


def test_argument_validation_error_codes():
    from mmgl_amd import _lib
    L = _lib.lib()
    # null pointers / bad sizes are rejected before any launch (safe without a GPU)
    assert L.mmgl_xattn_fwd(None, None, None, None, None, None, 1, 1, 8, 8, 64, 0, None) == 1
    assert b"null" in L.mmgl_last_error()
    assert L.mmgl_xattn_fwd(None, None, None, None, None, None, 1, 1, 8, 8, 48, 0, None) == 2     # head_dim
    assert L.mmgl_xattn_fwd(None, None, None, None, None, None, 1, 1, 8, 300, 64, 0, None) == 2   # S > 256
    assert L.mmgl_xattn_fwd(None, None, None, None, None, None, 0, 1, 8, 8, 64, 0, None) == 1
    assert L.mmgl_linear_fwd(None, None, None, None, 4, 4, 4, 0, 1.0, 1, None) == 1
    assert L.mmgl_xattn_bwd_workspace(4, 32, 640, 64, 64) > 4 * 32 * 640 * 4
    assert L.mmgl_linear_wgrad_workspace(44, 8192, 768, 1) >= (8192 + 768) * 48 * 2
    with pytest.raises(ValueError):
        _lib.check(2, "x")
    with pytest.raises(RuntimeError):
        _lib.check(3, "x")
    # the gradient-exchange entry points validate before they touch RCCL; the unique id needs no device
    assert L.mmgl_comm_init(2, 2, None, None) == 1 and L.mmgl_allreduce_sum(None, None, 4, 0, None) == 1
    assert L.mmgl_comm_destroy(None) == 0
    uid = (ctypes.c_char * 128)()
    rc = L.mmgl_comm_unique_id(uid)
    assert rc in (0, 2), L.mmgl_last_error()             # 2: no librccl.so on this machine
    assert rc != 0 or any(bytes(uid))


def test_no_cpu_fallback():
    from mmgl_amd import _lib, ops
    x = torch.randn(2, 4, 64)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.xattn_core(x, x, x, torch.ones(2, 4, dtype=torch.bool), 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.layer_norm(x, None, None)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.linear(x, torch.randn(8, 64))
    # a missing extension is an error, not a fallback
    saved, _lib._lib = _lib._lib, None
    path, _lib.LIB_PATH = _lib.LIB_PATH, "/nonexistent/libmmgl_hip.so"
    try:
        with pytest.raises(RuntimeError, match="not built"):
            _lib.lib()
    finally:
        _lib.LIB_PATH, _lib._lib = path, saved


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "mmgl_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
