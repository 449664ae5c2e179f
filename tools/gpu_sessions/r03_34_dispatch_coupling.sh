#!/bin/bash
# round 3, GPU session 34: does a nearly full XCD hold back the dispatch of a grid's workgroups to the other XCDs?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s34; mkdir -p $O
timeout 120 tools/ubench/ubench_dispatch_coupling 2>&1 | tee $O/ubench_dispatch_coupling.txt
