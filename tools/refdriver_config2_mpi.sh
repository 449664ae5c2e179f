#!/bin/bash
# BASELINE config 2 (32768^2, 23 x 23 blocks, 10 % fill, fp64) through the UNCHANGED reference as a multi-rank MPI program on this
# back end (oracle/_ref/host_acc_mpi), the ranks sharing the one GPU of the box: with N >= 4 ranks every rank's part of C fits the
# default integers the reference's device-memory layer counts bytes with, which one rank does not.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/refdriver_c2_mpi; mkdir -p $O
export MKL_THREADING_LAYER=SEQUENTIAL
python - <<'PY'
toks = [0, "F", "dbcsr_multiply", 32768, 32768, 32768, "0.9d0", "0.9d0", "0.9d0", "N", "N", "N", "N", "N", 3, "1.0d0", "0.0d0", "1.0d0", "0.0d0",
        0, 0, 0, 0, 0, 0, "F", 2, 1, 1, 1, 1, 23, 1, 23, 1, 23, "F", "0.1E-10", "0.0E+00", "0.0E+00"]
open("gpurun_out/refdriver_c2_mpi/config2.perf", "w").write("\n".join(str(t) for t in toks) + "\n")
PY
P=$PWD/$O/config2.perf
NC=$(nproc)
for n in ${NRANKS:-8 4}; do
  t=$(( NC / n )); [ $t -gt 16 ] && t=16; [ $t -lt 1 ] && t=1
  ( cd /tmp && OMP_NUM_THREADS=$t timeout 1200 ${MPIEXEC:-/opt/conda/bin/mpiexec} -n $n $OLDPWD/oracle/_ref/host_acc_mpi/dbcsr_perf $P > $OLDPWD/$O/acc_mpi_$n.txt 2>&1 ); echo "== host_acc_mpi, $n ranks x $t threads, rc $?"
  grep -E "numnodes \(|nthreads  |time  |perf total|flops total|matmuls total|checksum\(C_out\) " $O/acc_mpi_$n.txt
done
