cd /root/repo; mkdir -p gpurun_out
export PMC_SQ_COUNTERS="SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL"
echo "== TT"; bash tools/pmc_sq.sh "gemm8p" gpurun_out/lds_tt.txt -- python tools/probes/wgrad_one.py 40960 8192 2048 | tail -10
echo "== TT ablate 5 (b128)"; MMGL_LIB_PATH=build_probe/libmmgl_t8abl5.so bash tools/pmc_sq.sh "gemm8p" gpurun_out/lds_tt5.txt -- python tools/probes/wgrad_one.py 40960 8192 2048 | tail -10
echo "== NT"; bash tools/pmc_sq.sh "gemm8p" gpurun_out/lds_nt.txt -- python tools/probes/gemm8p_one.py 40960 2048 8192 | tail -10
tail -5 /tmp/pmc_sq.log
