#!/bin/bash
# round 6, GPU session 49: the k-pass threshold (1 MB of A per block row) at its edge: 31 / 32 blocks, 1425 block rows, fill 0.1 (1.1-1.2 MB per row: two passes) and
# fill 0.15 / 0.2 (three passes) -- whole step, automatic against one pass
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s49; mkdir -p $O; rm -f $O/*.jsonl
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B='[{"mix":"1,32","fill":0.1,"size":45600},{"mix":"1,31","fill":0.1,"size":44175},{"mix":"1,32","fill":0.15,"size":45600},{"mix":"1,32","fill":0.2,"size":32768},{"mix":"1,23","fill":0.2,"size":32775},{"mix":"1,28","fill":0.15,"size":39900}]'
for K in auto 1 2 3; do
  if [ $K = auto ]; then unset DBCSR_AMD_MM_KCHUNKS; else export DBCSR_AMD_MM_KCHUNKS=$K; fi
  timeout 900 python tools/block_bench.py --label k$K --batch "$B" 2>&1 | grep -v "$F" >> $O/k.jsonl
done
python3 - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r06_s49/k.jsonl") if l.startswith("{")]
for r in rows:
    if "error" in r: print(r); continue
    print("%-6s %-8s fill %.2f size %6d  passes %d  step_ms %8.3f  kernel_ms(one pass) %8.3f  TFLOP/s(step) %6.2f" % (r["label"], ",".join(map(str, r["mix_m"])), r["fill"], r["size"], r["k_passes"], r["ms_per_step"], r["kernel_ms"], r["tflops_step"]))
PY
