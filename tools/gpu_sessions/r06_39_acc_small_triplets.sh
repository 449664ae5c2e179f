#!/bin/bash
# round 6, GPU session 39: the acc ABI on the triplets of libsmm_acc's tiny dataflow (4 ... 8) and 9, 13: one stream in the reference timer's shape and sixteen streams
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s39; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
for S in 4 5 6 7 8 9 13; do
  timeout 120 python tools/acc_bench.py 100 16005 $S $S $S --check 2>&1 | grep -v "$F" | tail -1 >> $O/one_stream.txt
  timeout 120 python tools/acc_bench.py 30 30000 $S $S $S 4000 10000 10000 --threads 16 2>&1 | grep -v "$F" | tail -1 >> $O/sixteen_streams.txt
done
cat $O/one_stream.txt $O/sixteen_streams.txt | cut -c1-260
