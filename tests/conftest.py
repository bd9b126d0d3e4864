import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # bring libmmgl_hip.so up to date with its sources before anything imports it (a no-op when it is; a fresh checkout or an edited
    # kernel otherwise leaves the suite running against a missing / stale library).  Where hipcc is absent the prebuilt .so is used as is.
    from mmgl_amd import _build
    if os.path.exists(_build.HIPCC):
        try:
            _build.build(verbose=False)
        except Exception as e:             # keep a usable prebuilt library usable; tests/test_abi_cpu.py still fails on a stale one
            if not os.path.exists(_build.LIB):
                raise
            import warnings
            warnings.warn(f"could not rebuild libmmgl_hip.so ({e}); running against the existing one")


def pytest_collection_modifyitems(config, items):
    """-m gpu tests are skipped (not failed) if someone runs the whole suite on a CPU-only box."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
