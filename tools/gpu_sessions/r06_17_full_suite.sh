#!/bin/bash
# round 6, GPU session 17: the whole -m gpu suite after the day's kernel changes (padded pitch in the exact-size kernel, slab kernels, exact-size stack kernels,
# symmetric products on several ranks, the split of mm_engine.hip)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s17; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -25 ) > $O/pytest_gpu.txt 2>&1
tail -8 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -3
