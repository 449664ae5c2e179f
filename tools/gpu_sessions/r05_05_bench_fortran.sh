#!/bin/bash
# round 5, GPU session 5: the default bench line with this round's fields, the Fortran resident path (stamps, out-of-place products,
# product as operand) on 1 / 2 / 4 ranks, config 2 through the resident loop, counters of the fp32 kernels through bench.py's own passes
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r05_s05; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -4 $O/bench_default.err; cut -c1-1500 $O/bench_default.json
timeout 400 python -m pytest tests/test_gpu_fortran_host.py tests/test_fortran_host_mpi.py -q -k "out_of_place or keeps_matrices or stay_on_the_device" 2>&1 | grep -v "$F" | tail -15 > $O/pytest_fortran.txt
tail -5 $O/pytest_fortran.txt
export MKL_THREADING_LAYER=SEQUENTIAL OMP_NUM_THREADS=4 DBCSR_AMD_RESIDENT=0
for mode in 1 0 2; do
  echo "== one rank, config 2, mode $mode" >> $O/resident_loop_config2.txt
  timeout 200 oracle/_ref/host_resident/dbcsr_resident_loop 32768 0.9 23 8 0 $mode 2>&1 | grep "resident_loop" >> $O/resident_loop_config2.txt
done
echo "== two ranks sharing the GPU, config 2, mode 1" >> $O/resident_loop_config2.txt
timeout 240 /opt/conda/bin/mpiexec -n 2 oracle/_ref/host_resident_mpi/dbcsr_resident_loop 32768 0.9 23 6 0 1 2>&1 | grep "resident_loop" >> $O/resident_loop_config2.txt
cat $O/resident_loop_config2.txt
unset OMP_NUM_THREADS
for g in 0 4; do
  DBCSR_AMD_MM_F32_GROUP=$g timeout 300 python bench.py --workload fp32_16384_32x32_fill20 --no-other-configs --cpu-seconds 0 --steps 5 2>/dev/null | tail -1 >> $O/f32_counters.jsonl
done
cut -c1-1200 $O/f32_counters.jsonl
