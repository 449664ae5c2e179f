#!/bin/bash
# kernel time of every single-GPU workload with and without the exact-size kernels
for w in config1_4096_4x4_fill10_fp64 config2_32768_23x23_fill10_fp64 config3_32768_mixed13_23_32_fill5_fp64 config4_131072_23x23_fill1_fp64 mid_16384_23x23_fill10_fp64 mid_8192_23x23_fill10_fp64 small_4600_23x23_fill10_fp64 sparse_65536_23x23_fill1_fp64; do
  for h in 1 0; do
    echo "$w HOT=$h $(DBCSR_AMD_MM_HOT=$h python bench.py --workload $w --steps 5 --warmup 2 --cpu-seconds 0 2>&1 | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["roofline"]["kernel_ms"],3), round(j["value"]))' 2>&1 | tail -1)"
  done
done
