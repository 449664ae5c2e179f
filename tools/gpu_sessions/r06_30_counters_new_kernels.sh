#!/bin/bash
# round 6, GPU session 30: rocprofv3 kernel trace + counters (one pass per group, tools/profile_cmd.sh) of the kernels this round changed: hot<32,32,32> and
# hot<24,24,24> with the padded pitch, the one-wave slab kernels on 40^3 / 33^3 / 48 x 36 x 23
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s30; mkdir -p $O
B='[{"mix":"1,32","fill":0.1,"size":32768},{"mix":"1,24","fill":0.1,"size":32768},{"mix":"1,40","fill":0.2},{"mix":"1,33","fill":0.2},{"mix_m":"1,48","mix_n":"1,36","mix_k":"1,23","fill":0.2}]'
bash tools/profile_cmd.sh r06_new_kernels python $PWD/tools/block_bench.py --size 16384 --label counters --batch "$B" > $O/summary.txt 2>&1
cp gpurun_out/prof_r06_new_kernels/summary.txt $O/prof_summary.txt
tail -80 $O/prof_summary.txt | cut -c1-200
find gpurun_out/prof_r06_new_kernels -name "*.csv" -size +1M -delete
