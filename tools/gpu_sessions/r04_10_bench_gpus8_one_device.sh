#!/bin/bash
# round 4, GPU session 10: bench.py --gpus 8 end to end on the one-GPU box (ranks share the device over gloo): bounded probe, config 4 beside
# config 2, counter passes with the scaled timeout, parity checksum of the distributed result
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s10; mkdir -p $O
( time timeout 1500 python bench.py --gpus 8 --steps 5 --warmup 1 ) > $O/bench_gpus8.json 2> $O/bench_gpus8.err
tail -4 $O/bench_gpus8.err
python3 - $O/bench_gpus8.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        print({k: r.get(k) for k in ("value", "ms_per_step", "n_gpus", "multiplies_run", "parity_distributed_checksum_rel_err_vs_single_gpu", "parity_single_gpu_max_rel_err_vs_cpu_sample")})
        print(r["config"])
        print({k: v for k, v in r["roofline"].items() if k in ("kernel", "kernel_ms", "frac", "traffic", "mfma_busy_frac", "sclk_mhz", "l2_hit_rate")})
        print(r.get("workloads"))
        print(r.get("cpu_baseline", {}).get("value"))
PY
