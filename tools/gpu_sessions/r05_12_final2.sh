#!/bin/bash
# round 5, GPU session 12: the final tree once more (large-block kernels with exact tile counts, the acc ABI's single-variant instantiation):
# whole -m gpu suite, smoke(), acc_bench on the large blocks, the default bench line
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s12; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -30 ) > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -2 > $O/smoke.txt; cat $O/smoke.txt
for mnk in "80 80 80" "72 72 72" "64 64 64" "48 48 48" "40 40 40"; do
  timeout 120 python tools/acc_bench.py 5 16005 $mnk --check 2>&1 | grep acc_bench >> $O/acc_bench_final.txt
done
cat $O/acc_bench_final.txt
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err; cut -c1-500 $O/bench_default.json
