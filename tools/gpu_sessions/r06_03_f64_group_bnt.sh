#!/bin/bash
# round 6, GPU session 3: the fp64 group kernel with the non-temporal hint on its B loads (does a streamed B leave the A rows in L2?), R = 2 ... 6
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s03; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
B='[{"label":"hot","env":["DBCSR_AMD_MM_F64_GROUP=0"]}'
for R in 2 3 4 6; do B="$B"',{"label":"group_R'$R'_bnt","env":["DBCSR_AMD_MM_F64_GROUP='$R'","DBCSR_AMD_MM_GROUP_BNT=1"]}'; done
B="$B"']'
DBCSR_AMD_MM_GROUP_BNT=1 timeout 600 python tools/block_bench.py --size 32768 --mix 1,23 --fill 0.1 --steps 5 --check --batch "$B" 2>&1 | grep -v "$F" > $O/group_bnt.jsonl
python3 -c "
import json
for l in open('$O/group_bnt.jsonl'):
    if l.startswith('{'):
        d = json.loads(l); print(d.get('label'), d.get('kernel'), d.get('kernel_ms'), d.get('frac_of_peak_kernel'), d.get('check', {}).get('max_abs_diff_over_max_abs'), d.get('error'))
"
BENCH_ARGS="--no-other-configs" bash tools/pmc_quick.sh "TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" DBCSR_AMD_MM_GROUP_BNT=1 DBCSR_AMD_MM_F64_GROUP=4 2>&1 | tail -2
