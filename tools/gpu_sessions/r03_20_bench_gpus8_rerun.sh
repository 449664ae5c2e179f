#!/bin/bash
# round 3, GPU session 20: bench.py --gpus 8 / 4 from a bare shell on the one-GPU box after the counter-pass fix (four schedule candidates)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s20; mkdir -p $O
for n in 8 4; do
  SECONDS=0; timeout 1500 python bench.py --gpus $n --steps 3 --warmup 1 --workload mid_16384_23x23_fill10_fp64 --cpu-seconds 3 > $O/bench_gpus$n.json 2> $O/bench_gpus$n.err
  echo "gpus $n rc $? wall ${SECONDS}s"; tail -c 3000 $O/bench_gpus$n.json | head -c 2200; echo; grep -v "amdgpu.ids\|socket.cpp\|Gloo\|peer ranks\|^\s*$" $O/bench_gpus$n.err | tail -5
done
