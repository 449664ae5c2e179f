#!/bin/bash
# prints the message of a DBCSR_ABORT in the reference Fortran host (the reference does not flush its output before abort())
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export MKL_THREADING_LAYER=SEQUENTIAL OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
exec /opt/rocm/bin/rocgdb -batch -ex "break _QMdbcsr_base_hooksPdbcsr_abort" -ex run -ex "x/s \$rdi" -ex "x/s \$rsi" -ex bt --args "$@"
