#!/bin/bash
# round 5, GPU session 2: parity of the two new kernels (fp32 direct, fp64 blocks of 33 .. 80), their timings, config 5 with the direct
# kernel, two more 4-process walks of the randomised sweep, the fill sweep's dense end without k passes, issue priority in the 23^3 kernel
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s02; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 420 python -m pytest tests/test_gpu_big_blocks.py "tests/test_gpu_kernel_variants.py" -q -k "big or fp32 or VARIANT=5 or VARIANT=6 or BIG" 2>&1 | grep -v "$F" | tail -25 > $O/pytest_new_kernels.txt
tail -5 $O/pytest_new_kernels.txt
# (a) blocks of 33 .. 80 with the new kernel, each checked against the plain global-memory kernel on the device
B='[{"mix":"1,40","fill":0.2},{"mix":"1,64","fill":0.3},{"mix":"1,72","fill":0.3},{"mix":"1,80","fill":0.3},{"mix":"1,33","fill":0.2},{"mix":"1,55","fill":0.3},'
B="$B"'{"mix_m":"1,45","mix_n":"1,67","mix_k":"1,78","fill":0.3},{"mix":"1,72","fill":0.3,"env":["DBCSR_AMD_MM_KCHUNKS=1"],"label":"r05_big_one_pass"},{"mix":"1,72","fill":0.1,"size":32768,"label":"r05_big_32768_fill10"}]'
timeout 400 python tools/block_bench.py --size 16384 --label r05_big --check --batch "$B" 2>&1 | grep -v "$F" > $O/large_blocks_after.jsonl
# (b) fp32: direct against the staged kernel, cache-resident sizes
B='[{"size":16384},{"size":32768},{"size":16384,"env":["DBCSR_AMD_MM_F32_DIRECT=0"],"label":"staged"},{"size":32768,"env":["DBCSR_AMD_MM_F32_DIRECT=0"],"label":"staged"},'
B="$B"'{"size":32768,"env":["DBCSR_AMD_MM_WG_WAVES=1"],"label":"direct_ww1"},{"size":32768,"env":["DBCSR_AMD_MM_WG_WAVES=2"],"label":"direct_ww2"}]'
timeout 300 python tools/block_bench.py --mix 1,32 --fill 0.2 --dtype f32 --label direct --check --batch "$B" 2>&1 | grep -v "$F" > $O/f32_direct.jsonl
# (c) config 5 at full size: the direct kernel with 4 (default), 2 and 3 k passes, the staged kernel as it was
for spec in "direct_default:" "direct_k2:DBCSR_AMD_MM_KCHUNKS=2" "direct_k3:DBCSR_AMD_MM_KCHUNKS=3" "staged_default:DBCSR_AMD_MM_F32_DIRECT=0"; do
  L=${spec%%:*}; E=${spec#*:}
  ( [ -n "$E" ] && export $E; timeout 240 python -c "
import json, bench
r = bench.run_other_config('config5_131072_32x32_fill20_fp32')
r['label'] = '$L'
print(json.dumps(r))" 2>&1 | grep -v "$F" | tail -1 ) >> $O/config5.jsonl
done
# (d) two more 4-process walks of all 1680 cases
timeout 200 python tools/soak_multiproc.py --procs 4 --omp-threads 8 --budget-s 150 --out $O/soak_p4_t8_b > $O/soak_p4_t8_b.txt 2>&1
timeout 200 python tools/soak_multiproc.py --procs 4 --omp-threads 16 --budget-s 150 --out $O/soak_p4_t16 > $O/soak_p4_t16.txt 2>&1
# (e) fill sweep, dense end, one pass over k
B='['
for spec in "32768 0.4" "16384 0.8"; do
  set -- $spec
  B="$B{\"size\":$1,\"fill\":$2,\"label\":\"production_k1\"},{\"size\":$1,\"fill\":$2,\"lab\":true,\"env\":[\"DBCSR_AMD_MM_TILE=2\",\"DBCSR_AMD_MM_KCHUNKS=1\"],\"label\":\"tile_k1\"},"
  B="$B{\"size\":$1,\"fill\":$2,\"lab\":true,\"env\":[\"DBCSR_AMD_MM_BAND=2\",\"DBCSR_AMD_MM_KCHUNKS=1\"],\"label\":\"band_k1\"},"
done
B="$B{\"size\":32768,\"fill\":0.1,\"lab\":true,\"env\":[\"DBCSR_AMD_MM_HOT_VARIANT=5\"],\"label\":\"hot_setprio\"},{\"size\":32768,\"fill\":0.1,\"lab\":true,\"env\":[\"DBCSR_AMD_MM_HOT_VARIANT=6\"],\"label\":\"hot_setprio_unpaired\"},"
B="$B{\"size\":32768,\"fill\":0.1,\"lab\":true,\"label\":\"hot_lab_build\"},{\"size\":32768,\"fill\":0.4,\"lab\":true,\"env\":[\"DBCSR_AMD_MM_HOT_VARIANT=5\"],\"label\":\"hot_setprio\"}]"
timeout 400 env DBCSR_AMD_MM_KCHUNKS=1 python tools/block_bench.py --batch "$B" 2>&1 | grep -v "$F" > $O/fill_sweep_dense.jsonl
for f in $O/large_blocks_after.jsonl $O/f32_direct.jsonl $O/config5.jsonl $O/fill_sweep_dense.jsonl; do echo "== $f"; cut -c1-900 $f; done
for f in $O/soak_p4_t8_b.txt $O/soak_p4_t16.txt; do echo "== $f"; head -c 700 $f; echo; done
