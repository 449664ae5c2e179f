#!/bin/bash
# round 4, GPU session 7: N > 1 hardening -- native transport on a one-rank communicator, auto transport falling back, residency under a
# multi-rank Fortran host; config 2 through the resident loop on 1 / 2 / 4 ranks sharing the GPU
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s07; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_cannon_shared_gpu.py -k "native_transport or auto_transport" tests/test_fortran_host_mpi.py -k "native_transport or auto_transport or stay_on_the_device or rccl_request or resident_engine or falls_through" -x -q 2>&1 | tail -12 | tee $O/pytest.txt
timeout 600 python -m pytest tests/test_gpu_fortran_host.py -k "keeps_matrices" -x -q 2>&1 | tail -4 | tee -a $O/pytest.txt
export MKL_THREADING_LAYER=SEQUENTIAL OMP_NUM_THREADS=4 DBCSR_AMD_RESIDENT=0
for n in 1 2 4; do
  echo "== config 2 (32768, 23 x 23, 10 %), resident loop, $n rank(s) sharing the GPU" | tee -a $O/resident_loop_config2.txt
  timeout 900 mpiexec -n $n oracle/_ref/host_resident_mpi/dbcsr_resident_loop 32768 0.9 23 8 0 2>&1 | grep "resident_loop" | tee -a $O/resident_loop_config2.txt
done
