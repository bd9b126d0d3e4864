#!/usr/bin/env python
"""Build a VARIANT of libmmgl_hip.so for same-box A/B timing: one translation unit recompiled with extra -D flags, every other
object taken from the regular build.  Output: variants/lib_<name>.so (git-ignored, travels with gpurun; select it with
MMGL_LIB_PATH).   usage: python tools/build_variant.py <name> <file.hip> [-DFLAG=..]..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmgl_amd import _build  # noqa: E402


def main():
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    _build.build(verbose=False)
    out_dir = os.path.join(ROOT, "variants")
    os.makedirs(out_dir, exist_ok=True)
    base = os.path.basename(src)[:-4]
    obj = os.path.join(out_dir, f"{base}_{name}.o")
    subprocess.run([_build.HIPCC] + _build._cflags(src) + flags + ["-c", os.path.join(_build.CSRC, os.path.basename(src)), "-o", obj], check=True)
    objs = [obj if os.path.basename(o)[:-2] == base else o
            for o in (os.path.join(_build.OBJ, f) for f in sorted(os.listdir(_build.OBJ)) if f.endswith(".o"))]
    lib = os.path.join(out_dir, f"lib_{name}.so")
    subprocess.run([_build.HIPCC, "--offload-arch=" + _build.ARCH, "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(lib)


if __name__ == "__main__":
    main()
