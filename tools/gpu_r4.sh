#!/bin/bash
# round-4 GPU session script: bash tools/gpu_r4.sh <tag> <what...>
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for what in "$@"; do
case $what in
  tests_new)
    python -m pytest tests/test_rccl_gpu.py tests/test_gemm_nt_gpu.py tests/test_encoders_gpu.py tests/test_trainer_gpu.py -q -s 2>&1 | tail -60 > $out/tests_new.log;;
  tests_full)
    python -m pytest tests/test_model_gpu.py -q -s -k "full_size_step" 2>&1 | tail -40 > $out/tests_full.log;;
  tests_all)
    python -m pytest tests -m gpu -q 2>&1 | tail -40 > $out/tests_all.log;;
  bench_force)
    python -X faulthandler bench.py --force-exchange --no-cpu-baseline > $out/bench_force.json 2> $out/bench_force.err; echo "rc=$?" >> $out/bench_force.err;;
  bench_ab_dyn)
    python bench.py --no-cpu-baseline > $out/bench_static.json 2> $out/bench_static.err
    MMGL_GEMM_DYNAMIC=2 python bench.py --no-cpu-baseline > $out/bench_dyn.json 2> $out/bench_dyn.err
    python bench.py --no-cpu-baseline --config opt-125m > $out/bench125_static.json 2>> $out/bench_static.err
    MMGL_GEMM_DYNAMIC=2 python bench.py --no-cpu-baseline --config opt-125m > $out/bench125_dyn.json 2>> $out/bench_dyn.err;;
  xattn)
    python -m pytest tests/test_xattn_gpu.py tests/test_llama_gpu.py -q -x 2>&1 | tail -30 > $out/tests_xattn.log
    python tools/bench_xattn.py llama > $out/bench_xattn_llama.txt 2>&1
    python bench.py --config llama-2-7b --no-cpu-baseline > $out/bench_llama.json 2> $out/bench_llama.err;;
  xattn_pmc)
    bash tools/pmc_xattn.sh 8 llama > /dev/null 2>&1; cp gpurun_out/pmc_xattn_B8llama.txt $out/;;
  sa)
    timeout 900 python -m pytest tests/test_selfattn_gpu.py -q -x 2>&1 | tail -15 > $out/tests_sa.log
    timeout 600 bash tools/gpu_variants.sh python tools/bench_selfattn.py 64 > $out/bench_sa_variants.txt 2>&1
    timeout 600 bash tools/pmc_mem.sh "sa32" $out/pmc_sa32_mem.txt -- python tools/probes/selfattn_one.py 64 > /dev/null 2>&1;;
  llama)
    timeout 1200 python -m pytest tests/test_gemm_nt_gpu.py tests/test_llama_gpu.py -q -x 2>&1 | tail -8 > $out/tests_llama.log
    python bench.py --config llama-2-7b --no-cpu-baseline > $out/bench_llama.json 2> $out/bench_llama.err;;
  bench)
    python bench.py > $out/bench.json 2> $out/bench.err;;
  *) echo "unknown $what";;
esac
done
tail -3 $out/*.log 2>/dev/null
