#!/bin/bash
# round-2 GPU session 1: fabric micro-benchmark, parity + timing of the LDS-DMA exact-size kernel (S = 2, 3, 4) next to round 1's
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/s1; mkdir -p $O
timeout 300 ./tools/ubench/ubench_fabric > $O/ubench_fabric.txt 2>&1
for k in default dma2 dma3 dma4; do
  env DBCSR_AMD_MM_KERNEL=$k timeout 300 python -m pytest tests/test_gpu_multiply.py tests/test_gpu_native_multiply.py -x -q -m gpu > $O/pytest_$k.txt 2>&1
  if [ $k = default ]; then ( for i in $(seq 60); do rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|fclk|mclk" | head -4; sleep 0.3; done ) > $O/clocks.txt 2>&1 & fi
  env DBCSR_AMD_MM_KERNEL=$k timeout 300 python bench.py --steps 8 --warmup 1 --cpu-seconds 2 > $O/bench_$k.json 2> $O/bench_$k.err
  wait
done
tail -n 3 $O/pytest_*.txt
cat $O/bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('parity_max_rel_err_vs_cpu_sample'))
"
cat $O/ubench_fabric.txt
