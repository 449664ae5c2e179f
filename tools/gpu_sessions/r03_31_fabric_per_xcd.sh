#!/bin/bash
# round 3, GPU session 31: is the L2 <-> fabric ceiling a sum of per-XCD limits or one shared limit? (streaming read by the workgroups of 8 / 4 / 2 / 1 XCDs)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s31; mkdir -p $O
timeout 300 tools/ubench/ubench_fabric xcds 2>&1 | tee $O/ubench_fabric_xcds.txt
