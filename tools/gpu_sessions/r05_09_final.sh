#!/bin/bash
# round 5, GPU session 9: the final tree -- the whole -m gpu suite, smoke(), bench.py --gpus 8 on the one device (gloo), the default bench line
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s09; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -30 ) > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -2 > $O/smoke.txt; cat $O/smoke.txt
( time timeout 500 python bench.py --gpus 8 --steps 3 --warmup 1 --cpu-seconds 2 ) > $O/bench_gpus8_one_device.json 2> $O/bench_gpus8.err
tail -4 $O/bench_gpus8.err | cut -c1-200; grep '^{"metric"' $O/bench_gpus8_one_device.json | cut -c1-900
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err; cut -c1-500 $O/bench_default.json
