#!/bin/bash
# round 4, GPU session 4: band dataflow, shape 1 (16 waves per workgroup, 2 x 2 C blocks per wave, one A slot) -- parity, then windows and depths
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export DBCSR_AMD_LAB=1   # the band dataflow lives in the lab build (dbcsr_amd/csrc/Makefile)
O=gpurun_out/r04_s04; mkdir -p $O
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
  for w in 0 256 384 512 768 1024; do DBCSR_AMD_MM_BAND=2 DBCSR_AMD_MM_BAND_WINDOW=$w timeout 300 $B 2>&1 | line "band shape 1 window $w"; done
  for w in 0 384 768; do DBCSR_AMD_MM_BAND=2 DBCSR_AMD_MM_BAND_WINDOW=$w DBCSR_AMD_MM_BAND_KNOBS=1 timeout 300 $B 2>&1 | line "band shape 1 window $w timing"; done
  for d in 12 16 22; do DBCSR_AMD_MM_BAND=2 DBCSR_AMD_MM_BAND_DEPTH=$d timeout 300 $B 2>&1 | line "band shape 1 window 384 depth $d"; done
  DBCSR_AMD_MM_BAND=2 DBCSR_AMD_MM_BAND_BPOL=1 timeout 300 $B 2>&1 | line "band shape 1 window 384 nt"
  DBCSR_AMD_MM_BAND=2 DBCSR_AMD_MM_BAND_SHAPE=0 timeout 300 $B 2>&1 | line "band shape 0 window 384"
  ) | tee $O/bench_lines.txt
for w in 384 768; do
( tools/pmc_quick.sh "FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" DBCSR_AMD_MM_BAND=2 DBCSR_AMD_MM_BAND_WINDOW=$w
  tools/pmc_quick.sh "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" DBCSR_AMD_MM_BAND=2 DBCSR_AMD_MM_BAND_WINDOW=$w ) 2>&1 | tee -a $O/pmc.txt
done
