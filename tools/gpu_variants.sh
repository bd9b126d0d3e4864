#!/bin/bash
# same-box A/B of library variants (tools/build_variant.py): runs "$@" once per variants/lib_*.so and once with the regular build
echo "== base"; "$@" 2>&1 | grep -v amdgpu.ids
for f in variants/lib_*.so; do echo "== $f"; MMGL_LIB_PATH=$PWD/$f "$@" 2>&1 | grep -v amdgpu.ids; done
