"""Build libmmgl_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m mmgl_amd._build [--force]

One translation unit per kernel family under mmgl_amd/csrc/*.hip, compiled in parallel, linked into
mmgl_amd/libmmgl_hip.so.  The .so is git-ignored but travels to the GPU box with the tree.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmmgl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-return-type",
          "-fno-gpu-rdc", "-munsafe-fp-atomics",
          # MFMA accumulators in plain VGPRs: the attention kernels do softmax arithmetic on every accumulator between MFMAs;
          # with the default AGPR form a quarter of their inner-loop instructions were v_accvgpr_read/write copies
          "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _cflags(src):
    return list(CFLAGS)


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "mmgl_hip.h")]
    extra = os.environ.get("MMGL_HIPCC_FLAGS", "").split()
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or extra or _newer(s, o, deps):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + _cflags(s) + extra + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[mmgl build] compiled {os.path.basename(s)}", flush=True)
        return o

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[mmgl build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
