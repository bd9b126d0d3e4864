"""mmgl_amd: MI355X-native (gfx950) implementation of MMGL's neighbor-fusion hot path.

Host side mirrors the reference's Python module API (model.CrossAttentionModel / SelfAttentionModel,
language_modelling.run_generation.Arguments, wikiweb2m.WikiWeb2M); compute goes through hand-written HIP
kernels in libmmgl_hip.so (C ABI: include/mmgl_hip.h) loaded with ctypes.
"""
__version__ = "0.1.0"
