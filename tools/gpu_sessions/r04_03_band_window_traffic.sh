#!/bin/bash
# round 4, GPU session 3: band dataflow, fabric bytes and L2 hit rate against the width of the XCD's k window
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export DBCSR_AMD_LAB=1   # the band dataflow lives in the lab build (dbcsr_amd/csrc/Makefile)
O=gpurun_out/r04_s03; mkdir -p $O
for w in 192 256 384 512 768 1024 2048; do
( tools/pmc_quick.sh "FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" DBCSR_AMD_MM_BAND=2 DBCSR_AMD_MM_BAND_WINDOW=$w
  tools/pmc_quick.sh "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" DBCSR_AMD_MM_BAND=2 DBCSR_AMD_MM_BAND_WINDOW=$w ) 2>&1 | tee -a $O/pmc.txt
done
