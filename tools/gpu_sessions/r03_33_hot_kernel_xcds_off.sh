#!/bin/bash
# round 3, GPU session 33: the exact-size kernel with whole XCDs switched off (their C blocks are not computed: timing only).  Does an XCD run
# faster when its neighbours leave the fabric alone?  Kernel time = time of the slowest active XCD, each doing its usual eighth of the work.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s33; mkdir -p $O
for m in 0x00 0x01 0x11 0x33 0x0f 0x55 0x77 0xfe; do
  dbg=$(( (m << 8) | 0x10000 ))
  DBCSR_AMD_MM_DBG=$dbg timeout 300 python bench.py --steps 6 --warmup 2 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
m=int('$m',16); act=8-bin(m).count('1')
print('XCDs off mask $m  active %d  kernel %.3f ms  (full-chip equivalent at this per-XCD speed: %.3f ms)' % (act, r['kernel_ms'], r['kernel_ms']))"
done | tee $O/hot_xcds_off.txt
