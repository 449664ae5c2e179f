#!/bin/bash
# round 6, GPU session 34: bench.py --gpus 4 / 8 on the box's ONE device again (the ranks time-slice it; gloo, host-staged exchange), the schedule probe now with tilepipe;
# and bench.py --gpus 4 --dist-mode tilepipe: the whole bench line with that schedule (distributed check against the one-GPU product included)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s34; mkdir -p $O
for N in 4 8; do
( time timeout 900 python bench.py --gpus $N --steps 3 --warmup 1 --cpu-seconds 2 ) > $O/bench_gpus${N}_one_device.json 2> $O/bench_gpus$N.err
tail -3 $O/bench_gpus$N.err | cut -c1-200; grep '^{"metric"' $O/bench_gpus${N}_one_device.json | cut -c1-1800
done
( time timeout 900 python bench.py --gpus 4 --steps 3 --warmup 1 --cpu-seconds 2 --dist-mode tilepipe ) > $O/bench_gpus4_tilepipe.json 2> $O/bench_gpus4_tilepipe.err
tail -3 $O/bench_gpus4_tilepipe.err | cut -c1-200; grep '^{"metric"' $O/bench_gpus4_tilepipe.json | cut -c1-1800
