#!/bin/bash
# config 4 ablations of the exact-size kernel (DBCSR_AMD_MM_DBG: 1 no operand loads, 2 no MFMAs, 4 no LDS copies, 16 plain instead of streaming stores)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/s27; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernel_variants.py -x -q -m gpu 2>&1 | tail -2
for d in 0 1 2 3 7 16; do
echo "dbg=$d $(DBCSR_AMD_MM_DBG=$d timeout 300 python bench.py --workload config4_131072_23x23_fill1_fp64 --steps 4 --warmup 1 --cpu-seconds 0 --no-pmc 2>&1 | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["roofline"]["kernel_ms"],3), round(j["roofline"]["fill_products_ms"],3))')"
done | tee $O/ablate_c4.txt
