#!/bin/bash
# round 5, GPU session 3: parity tests of the new kernels (session 2's -k expression was malformed: nothing ran), counters of the fp32
# direct kernel (what bounds it: LDS is not it), k passes at 20 / 30 % fill, the default-thread walk after the oracle's team got bounded
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s03; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 500 python -m pytest tests/test_gpu_big_blocks.py tests/test_gpu_kernel_variants.py tests/test_gpu_native_multiply.py tests/test_gpu_multiply.py -q -k "big or fp32 or VARIANT or k_pass or kpass or f32" 2>&1 | grep -v "$F" | tail -25 > $O/pytest_new_kernels.txt
tail -5 $O/pytest_new_kernels.txt
bash tools/profile_cmd.sh r05_f32_direct_32768 python tools/block_bench.py --size 32768 --mix 1,32 --fill 0.2 --dtype f32 --steps 2 > $O/prof_f32_direct.txt 2>&1
cp gpurun_out/prof_r05_f32_direct_32768/summary.txt $O/prof_f32_direct_summary.txt
B='[{"size":32768,"fill":0.2,"env":["DBCSR_AMD_MM_KCHUNKS=2"],"label":"production_k2"},{"size":32768,"fill":0.3,"label":"production_auto"},{"size":32768,"fill":0.3,"env":["DBCSR_AMD_MM_KCHUNKS=2"],"label":"production_k2"},'
B="$B"'{"size":32768,"fill":0.3,"env":["DBCSR_AMD_MM_KCHUNKS=3"],"label":"production_k3"},{"size":32768,"fill":0.3,"lab":true,"env":["DBCSR_AMD_MM_BAND=2","DBCSR_AMD_MM_KCHUNKS=1"],"label":"band_k1"},'
B="$B"'{"size":16384,"mix":"1,72","fill":0.3,"label":"big72_auto_passes"},{"size":16384,"mix":"1,40","fill":0.3,"label":"big40"},{"size":16384,"mix":"1,64","fill":0.3,"env":["DBCSR_AMD_MM_KCHUNKS=1"],"label":"big64_k1"}]'
timeout 300 python tools/block_bench.py --batch "$B" 2>&1 | grep -v "$F" > $O/sweeps.jsonl
timeout 200 python tools/soak_multiproc.py --procs 4 --omp-threads 0 --budget-s 150 --out $O/soak_p4_default_threads > $O/soak_p4_default_threads.txt 2>&1
cut -c1-700 $O/sweeps.jsonl; head -c 700 $O/soak_p4_default_threads.txt; echo; tail -60 $O/prof_f32_direct_summary.txt
