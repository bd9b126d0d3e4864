python tools/probes/gemm_stagger.py
for r in 10 15; do for c in 12288 20480 28672; do for g in 2 4; do
MMGL_GEMM_STAGGER=$c MMGL_GEMM_STAGGER_GROUPS=$g MMGL_GEMM_STAGGER_ROUNDS=$r timeout 120 python tools/probes/gemm_stagger.py
done; done; done
python tools/probes/gemm_stagger.py
