#!/bin/bash
# round 5, GPU session 1: (a) the multi-process parity walk with per-case records (tools/soak_multiproc.py), (b) fill sweep of the 23 x 23
# path: production against the lab's tile / band dataflows, (c) where blocks of 33 .. 80 and the fp32 path stand before this round's kernels
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s01; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
nproc > $O/host.txt; free -g | head -2 >> $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>/dev/null
# (0) the refactored sweep file still says what it said
timeout 200 python -m pytest tests/test_gpu_random_sweep.py -q -x 2>&1 | grep -v "$F" | tail -3 > $O/pytest_sweep.txt
# (a) walks: 4 x 8 threads, 8 x 4 threads (the GPU side of the multi-process hypothesis), then 4 x host default (round 4's session 18 mode)
timeout 210 python tools/soak_multiproc.py --procs 4 --omp-threads 8 --budget-s 150 --out $O/soak_p4_t8 > $O/soak_p4_t8.txt 2>&1
timeout 180 python tools/soak_multiproc.py --procs 8 --omp-threads 4 --budget-s 120 --out $O/soak_p8_t4 > $O/soak_p8_t4.txt 2>&1
timeout 150 python tools/soak_multiproc.py --procs 4 --omp-threads 0 --budget-s 90 --out $O/soak_p4_t0 > $O/soak_p4_t0.txt 2>&1
# (b) fill sweep (one process)
B='['
for spec in "32768 0.1" "32768 0.2" "32768 0.4" "16384 0.8"; do
  set -- $spec
  B="$B{\"size\":$1,\"fill\":$2,\"label\":\"production\"},{\"size\":$1,\"fill\":$2,\"lab\":true,\"env\":[\"DBCSR_AMD_MM_TILE=2\"],\"label\":\"tile\"},"
  B="$B{\"size\":$1,\"fill\":$2,\"lab\":true,\"env\":[\"DBCSR_AMD_MM_BAND=2\"],\"label\":\"band_shape1_w384\"},"
done
B="$B{\"size\":32768,\"fill\":0.05,\"label\":\"production\"}]"
timeout 420 python tools/block_bench.py --batch "$B" 2>&1 | grep -v "$F" > $O/fill_sweep.jsonl
# (c) blocks of 33 .. 80 and fp32 as they stand
B='[{"mix":"1,40","fill":0.2},{"mix":"1,64","fill":0.3},{"mix":"1,72","fill":0.3},{"mix":"1,80","fill":0.3},{"mix":"1,33","fill":0.2},{"mix":"1,55","fill":0.3},'
B="$B"'{"mix_m":"1,45","mix_n":"1,67","mix_k":"1,78","fill":0.3},{"mix":"1,32","fill":0.2,"dtype":"f32"},{"mix":"1,32","fill":0.2,"dtype":"f32","size":32768}]'
timeout 300 python tools/block_bench.py --size 16384 --label r04_kernel --batch "$B" 2>&1 | grep -v "$F" > $O/blocks_before.jsonl
cat $O/pytest_sweep.txt; tail -2 $O/soak_p4_t8.txt $O/soak_p8_t4.txt $O/soak_p4_t0.txt; cat $O/fill_sweep.jsonl $O/blocks_before.jsonl | cut -c1-700
