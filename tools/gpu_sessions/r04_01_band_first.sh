#!/bin/bash
# round 4, GPU session 1: the band dataflow (mm_band.h) -- parity, then config 2 against the production kernel on the same box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export DBCSR_AMD_LAB=1   # the band dataflow lives in the lab build (dbcsr_amd/csrc/Makefile)
O=gpurun_out/r04_s01; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_band_kernel.py -x -q 2>&1 | tail -15 | tee $O/pytest.txt
B="python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-pmc"
line() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); f=r['roofline']; print('$1', 'ms_per_step %.3f kernel_ms %.3f frac %.4f kernel %s' % (r['ms_per_step'], f['kernel_ms'], f['frac'], f['kernel']))
    elif 'band kernel' in l: print(l.strip())
"; }
( timeout 300 $B 2>&1 | line "production"
  for d in 20 12 16 22; do DBCSR_AMD_MM_BAND=2 DBCSR_AMD_MM_BAND_DEPTH=$d timeout 300 $B 2>&1 | line "band depth $d"; done
  DBCSR_AMD_MM_BAND=2 DBCSR_AMD_MM_BAND_BPOL=1 timeout 300 $B 2>&1 | line "band depth 20 nt"
  DBCSR_AMD_MM_BAND=2 DBCSR_AMD_MM_BAND_KNOBS=1 timeout 300 $B 2>&1 | line "band depth 20 timing"
  timeout 300 $B 2>&1 | line "production again" ) | tee $O/bench_lines.txt
( tools/pmc_quick.sh "FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" DBCSR_AMD_MM_BAND=2
  tools/pmc_quick.sh "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" DBCSR_AMD_MM_BAND=2
  tools/pmc_quick.sh "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" DBCSR_AMD_MM_BAND=2 ) 2>&1 | tee $O/pmc.txt
