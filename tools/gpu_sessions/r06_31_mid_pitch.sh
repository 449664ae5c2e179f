#!/bin/bash
# round 6, GPU session 31: the slab kernels with conflict-free slab pitches (A's columns 48 doubles apart through 8-byte LDS writes, B's KSL + 2): parity, then
# the block sizes of sessions 10 / 24 / 28 again (33^3 was 12.33 ms = 0.364 of the fp64 peak, 40^3 8.13 = 0.553; SQ_LDS_BANK_CONFLICT 32 % / 19 % of the LDS cycles)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s31; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( timeout 900 python -m pytest tests/test_gpu_big_blocks.py tests/test_gpu_libsmm.py tests/test_gpu_kernel_variants.py -q -x 2>&1 | grep -v "$F" | tail -5 ) > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
DBCSR_AMD_SWEEP_FORCED=60 DBCSR_AMD_SWEEP_PLAIN=100 DBCSR_AMD_SWEEP_BIG=200 DBCSR_AMD_SWEEP_MID=600 timeout 900 python -m pytest tests/test_gpu_random_sweep.py -q -x -n 4 2>&1 | grep -v "$F" | tail -3
B='[{"mix":"1,33","fill":0.2},{"mix":"1,36","fill":0.2},{"mix":"1,37","fill":0.2},{"mix":"1,40","fill":0.2},{"mix":"1,44","fill":0.2},{"mix":"1,48","fill":0.2},{"mix_m":"1,48","mix_n":"1,36","mix_k":"1,23","fill":0.2},{"mix_m":"1,36","mix_n":"1,40","mix_k":"1,23","fill":0.2}]'
timeout 400 python tools/block_bench.py --size 16384 --label pitch48 --check --batch "$B" 2>&1 | grep -v "$F" > $O/slab.jsonl
python3 - <<'PY'
import json
for l in open("gpurun_out/r06_s31/slab.jsonl"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["label"], d.get("mix_m"), d.get("mix_n"), d.get("mix_k"), d.get("fill"), d.get("kernel"), "kernel_ms", d.get("kernel_ms"), "frac", d.get("frac_of_peak_kernel"), "diff", (d.get("check") or {}).get("max_abs_diff_over_max_abs"), d.get("error"))
PY
for mnk in "33 33 33" "36 36 36" "40 40 40"; do
  timeout 200 python tools/acc_bench.py 10 30000 $mnk 2000 400 400 --threads 16 2>&1 | grep acc_bench | cut -c1-200 >> $O/acc_bench.txt
done
cat $O/acc_bench.txt
( export DBCSR_AMD_MM_VERBOSE=0; timeout 300 python -c "
import json, bench
r = bench.run_other_config('config3_32768_mixed13_23_32_fill5_fp64', steps=5)
print('config3', r.get('ms_per_step'), r.get('value'))" 2>/dev/null | grep config3 )
