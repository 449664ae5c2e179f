#!/bin/bash
# round 6, GPU session 22: bench.py --gpus 4 / 8 on the box's ONE device (the ranks time-slice it; gloo, host-staged exchange): the whole N-rank code path
# of the bench with the schedule probe (colpipe, colpipe2d, gather, ticks), the distributed check of config 2 against the one-GPU product, the shared-GPU
# Cannon tests with the new schedule
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s22; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests/test_gpu_cannon_shared_gpu.py -q -x -k "colpipe2d" 2>&1 | grep -v "$F" | tail -4
for N in 4 8; do
( time timeout 900 python bench.py --gpus $N --steps 3 --warmup 1 --cpu-seconds 2 ) > $O/bench_gpus${N}_one_device.json 2> $O/bench_gpus$N.err
tail -3 $O/bench_gpus$N.err | cut -c1-200; grep '^{"metric"' $O/bench_gpus${N}_one_device.json | cut -c1-1500
done
