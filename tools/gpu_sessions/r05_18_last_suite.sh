#!/bin/bash
# round 5, GPU session 18: the whole -m gpu suite and smoke() on the last tree of the round
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s18; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 560 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -30 ) > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
