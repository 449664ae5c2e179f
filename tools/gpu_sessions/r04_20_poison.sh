#!/bin/bash
# round 4, GPU session 20: the randomised sweep on poisoned memory (every work area of the engine and every torch.empty filled with 0xFF / 0xCD first)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s20; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_poison.py -q -rf --tb=short 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -150 > $O/poison.txt; tail -120 $O/poison.txt
