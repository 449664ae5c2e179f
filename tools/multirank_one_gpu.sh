#!/bin/bash
# multi-rank paths on the one GPU of the box: Cannon driver incl. device-resident distributed input, comm ABI, bench.py's N > 1 code path (gloo)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/s15; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_cannon_shared_gpu.py tests/test_gpu_comm.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for n in ${NRANKS:-2 4}; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 3 --warmup 1 \
  --dist-backend gloo --workload config2_32768_23x23_fill10_fp64 --cpu-seconds 0 > $O/bench_gloo_$n.json 2> $O/bench_gloo_$n.err
echo "rc=$?"; tail -c 1500 $O/bench_gloo_$n.json; tail -3 $O/bench_gloo_$n.err
done
