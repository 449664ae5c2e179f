#!/bin/bash
# round 5, GPU session 11: the large-block kernels with every wave multiplying exactly the tiles it owns (against all TM x TN issued), parity;
# config 5 with the counter passes of bench.py (operands released before the passes: the child needs the device's memory)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s11; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 400 python -m pytest tests/test_gpu_big_blocks.py tests/test_gpu_libsmm.py tests/test_gpu_multiply.py -q -k "big or libsmm or validate or stack or h2o_like_80 or large" 2>&1 | grep -v "$F" | tail -8 > $O/pytest.txt
tail -4 $O/pytest.txt
B='['
for spec in "1,72 0.3" "1,40 0.2" "1,55 0.3" "1,33 0.2" "1,64 0.3" "1,80 0.3"; do
  set -- $spec
  B="$B{\"mix\":\"$1\",\"fill\":$2,\"label\":\"exact_tiles\"},{\"mix\":\"$1\",\"fill\":$2,\"env\":[\"DBCSR_AMD_MM_BIG=2\"],\"label\":\"all_tiles\"},"
done
B="$B{\"mix_m\":\"1,45\",\"mix_n\":\"1,67\",\"mix_k\":\"1,78\",\"fill\":0.3,\"label\":\"exact_tiles\"},{\"mix_m\":\"1,45\",\"mix_n\":\"1,67\",\"mix_k\":\"1,78\",\"fill\":0.3,\"env\":[\"DBCSR_AMD_MM_BIG=2\"],\"label\":\"all_tiles\"}]"
timeout 400 python tools/block_bench.py --size 16384 --check --batch "$B" 2>&1 | grep -v "$F" > $O/large_blocks_exact.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r05_s11/large_blocks_exact.jsonl"):
    if l.startswith("{"):
        r = json.loads(l)
        print(r.get("label"), r.get("mix_m"), r.get("mix_n"), r.get("kernel"), r.get("kernel_ms"), r.get("tflops_kernel"), r.get("frac_of_peak_kernel"), (r.get("check") or {}).get("max_abs_diff_over_max_abs"), r.get("error"))
PY
for mnk in "72 72 72" "40 40 40" "55 55 55" "45 67 78" "33 33 33" "80 80 80"; do
  timeout 120 python tools/acc_bench.py 5 16005 $mnk --check 2>&1 | grep acc_bench | sed 's/^/[exact tiles] /' >> $O/acc_bench_exact.txt
  DBCSR_AMD_SMM_BIG_EXACT=0 timeout 120 python tools/acc_bench.py 5 16005 $mnk 2>&1 | grep acc_bench | sed 's/^/[all tiles]   /' >> $O/acc_bench_exact.txt
done
cat $O/acc_bench_exact.txt
( time timeout 900 python bench.py --workload config5_131072_32x32_fill20_fp32 --steps 2 --warmup 1 --cpu-seconds 0 --no-other-configs ) > $O/bench_config5.json 2> $O/bench_config5.err
tail -3 $O/bench_config5.err | cut -c1-200; cut -c1-1500 $O/bench_config5.json
