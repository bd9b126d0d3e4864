mkdir -p gpurun_out
{
python tools/probes/gemm_stagger.py
for g in 4 8; do for c in 1024 2048 3072 4096 6144 8192 12288; do
MMGL_GEMM_STAGGER=$c MMGL_GEMM_STAGGER_GROUPS=$g timeout 120 python tools/probes/gemm_stagger.py
done; done
MMGL_GEMM_STAGGER=2048 MMGL_GEMM_STAGGER_GROUPS=16 timeout 120 python tools/probes/gemm_stagger.py
MMGL_GEMM_STAGGER=1024 MMGL_GEMM_STAGGER_GROUPS=32 timeout 120 python tools/probes/gemm_stagger.py
MMGL_GEMM_STAGGER=20480 MMGL_GEMM_STAGGER_GROUPS=4 timeout 120 python tools/probes/gemm_stagger.py
python tools/probes/gemm_stagger.py
} 2>&1 | tee gpurun_out/stagger.log | cut -c1-330
