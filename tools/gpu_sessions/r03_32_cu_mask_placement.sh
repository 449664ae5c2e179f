#!/bin/bash
# round 3, GPU session 32: placement of workgroups under hipExtStreamCreateWithCUMask (which mask bits are which XCD?)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s32; mkdir -p $O
timeout 120 tools/ubench/ubench_cu_mask 2>&1 | tee $O/ubench_cu_mask.txt
