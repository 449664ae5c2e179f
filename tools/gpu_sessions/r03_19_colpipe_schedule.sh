#!/bin/bash
# round 3, GPU session 19: the colpipe schedule (N x 1 grid, B in column chunks) on the HIP engine with ranks sharing the GPU; bench.py --gpus N flow with it
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r03_s19; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_cannon_shared_gpu.py -q -m gpu -x -k "colpipe" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for n in 2 8; do
  SECONDS=0; timeout 1500 python bench.py --gpus $n --steps 3 --warmup 1 --workload mid_16384_23x23_fill10_fp64 --cpu-seconds 3 > $O/bench_gpus$n.json 2> $O/bench_gpus$n.err
  echo "gpus $n rc $? wall ${SECONDS}s"; tail -c 1800 $O/bench_gpus$n.json | head -c 900; echo; grep -v "amdgpu.ids\|socket.cpp\|Gloo\|^\s*$" $O/bench_gpus$n.err | tail -5
done
SECONDS=0; timeout 900 python bench.py --gpus 4 --steps 3 --warmup 1 --workload mid_16384_23x23_fill10_fp64 --cpu-seconds 0 --no-pmc --dist-mode colpipe --col-chunks 6 > $O/bench_gpus4_colpipe.json 2> $O/bench_gpus4_colpipe.err
echo "gpus 4 colpipe rc $? wall ${SECONDS}s"; tail -c 1500 $O/bench_gpus4_colpipe.json | head -c 700; echo; grep -v "amdgpu.ids\|socket.cpp\|Gloo\|^\s*$" $O/bench_gpus4_colpipe.err | tail -5
