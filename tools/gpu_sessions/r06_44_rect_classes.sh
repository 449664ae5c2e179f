#!/bin/bash
# round 6, GPU session 44: uniform rectangular triplets through the class kernels (one class, one inner size): parity, then session 43's sweep again
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s44; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1200 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_multiply.py tests/test_gpu_small_blocks.py tests/test_gpu_filter.py -q -m gpu -x 2>&1 | grep -v "$F" | tail -4 | tee $O/pytest.txt
DBCSR_AMD_SWEEP_PLAIN=600 DBCSR_AMD_SWEEP_FORCED=300 timeout 900 python -m pytest tests/test_gpu_random_sweep.py -q -m gpu -x -n 4 -k "matches_oracle or forced" 2>&1 | grep -v "$F" | tail -3 | tee -a $O/pytest.txt
sed -e 's#gpurun_out/r06_s43#gpurun_out/r06_s44#g' -e 's#O=gpurun_out/r06_s43#O=gpurun_out/r06_s44#' tools/gpu_sessions/r06_43_rect_sweep.sh | sed -n '/^B=/,$p' > /tmp/rect44.sh
O=gpurun_out/r06_s44 F="$F" bash /tmp/rect44.sh
