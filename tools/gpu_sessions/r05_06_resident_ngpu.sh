#!/bin/bash
# round 5, GPU session 6: config 2 through the Fortran resident loop after the k-pass probe stopped costing the plan; the N-rank bench line
# (two ranks on the one device, gloo) with the roofline summed over a step's launches; the acc ABI on blocks of 33 .. 80
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s06; mkdir -p $O
export MKL_THREADING_LAYER=SEQUENTIAL
for mode in 1 0; do
  echo "== one rank, config 2, mode $mode" >> $O/resident_loop_config2.txt
  OMP_NUM_THREADS=4 DBCSR_AMD_RESIDENT=0 timeout 200 oracle/_ref/host_resident/dbcsr_resident_loop 32768 0.9 23 8 0 $mode 2>&1 | grep "resident_loop" >> $O/resident_loop_config2.txt
done
for n in 2 4; do
  echo "== $n ranks sharing the GPU, config 2, mode 1" >> $O/resident_loop_config2.txt
  OMP_NUM_THREADS=4 DBCSR_AMD_RESIDENT=0 timeout 240 /opt/conda/bin/mpiexec -n $n oracle/_ref/host_resident_mpi/dbcsr_resident_loop 32768 0.9 23 6 0 1 2>&1 | grep "resident_loop" >> $O/resident_loop_config2.txt
done
echo "== 2 ranks, M 8192, mode 2 (product -> operand on two ranks)" >> $O/resident_loop_config2.txt
OMP_NUM_THREADS=4 DBCSR_AMD_RESIDENT=0 timeout 240 /opt/conda/bin/mpiexec -n 2 oracle/_ref/host_resident_mpi/dbcsr_resident_loop 8192 0.9 23 4 1 2 2>&1 | grep "resident_loop" >> $O/resident_loop_config2.txt
grep -v "GFLOP/s of\|blocks of C" $O/resident_loop_config2.txt
( time timeout 420 python bench.py --gpus 2 --steps 5 --warmup 1 --cpu-seconds 2 --no-other-configs ) > $O/bench_gpus2_one_device.json 2> $O/bench_gpus2.err
tail -5 $O/bench_gpus2.err | cut -c1-300; cut -c1-2500 $O/bench_gpus2_one_device.json
for mnk in "40 40 40" "72 72 72" "45 67 78" "64 64 64" "23 23 23"; do
  timeout 120 python tools/acc_bench.py 5 16005 $mnk --check 2>&1 | grep acc_bench >> $O/acc_bench_blocks.txt
done
cat $O/acc_bench_blocks.txt
