#!/bin/bash
# Builds, from the reference sources WHERE THEY LIE under /root/reference (nothing is copied
# into the repo), the two pieces of the reference that compile from their own few files:
#   oracle/_ref/dbcsr_acc_test    the reference's C-ABI specification test
#                                 (tests/dbcsr_acc_test.c, pure C, only needs acc.h) linked
#                                 against THIS repo's libdbcsr_acc_amd.so
#   oracle/_ref/ref_stack_driver  the reference's kernel-validator host functions
#                                 (src/acc/libsmm_acc/libsmm_acc_benchmark.cpp: matInit,
#                                 stackInit, stackCalc, stackTransp, checkSum...) behind a small
#                                 driver of ours; used to validate the oracle's restatement.
# The full library (Fortran + fypp + generated parameters.h/smm_acc_kernels.h) is NOT
# buildable in this image and is not attempted.
set -e
REF=/root/reference
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
[ -d "$REF" ] || { echo "no reference checkout: keeping prebuilt oracle/_ref"; exit 0; }
mkdir -p "$OUT"
gcc -O2 -fopenmp -DNDEBUG -I"$REF/src" "$REF/tests/dbcsr_acc_test.c" -L"$HERE/../dbcsr_amd" -ldbcsr_acc_amd \
    -Wl,-rpath,'$ORIGIN/../../dbcsr_amd' -o "$OUT/dbcsr_acc_test"
g++ -O2 -D__HIP -D__HIP_PLATFORM_AMD__ -ffunction-sections -fdata-sections -I/opt/rocm/include -I"$REF/src/acc/libsmm_acc" \
    -c "$REF/src/acc/libsmm_acc/libsmm_acc_benchmark.cpp" -o "$OUT/libsmm_acc_benchmark.o"
g++ -O2 "$HERE/ref_stack_driver.cpp" "$OUT/libsmm_acc_benchmark.o" -Wl,--gc-sections -L/opt/rocm/lib -lamdhip64 \
    -Wl,-rpath,/opt/rocm/lib -o "$OUT/ref_stack_driver"
rm -f "$OUT/libsmm_acc_benchmark.o"
# Fortran host check: the reference's own device-binding module (src/acc/dbcsr_acc_device.F + the base modules it
# uses; these need no fypp) compiled with amdflang where they lie, linked with tests/fortran/dbcsr_amd_host_check.F90
# and THIS repo's library.  (dbcsr_acc_stream/event/init/devmem pull in dbcsr_config -> dbcsr_mpiwrap -> fypp and are
# not buildable here.)
if command -v amdflang >/dev/null 2>&1; then
  TMPF="$(mktemp -d)"
  FFLAGS="-cpp -ffree-form -O1 -D__DBCSR_ACC -D__HIP -I$REF/src -I$REF/src/base -I$TMPF -module-dir $TMPF"
  for f in base/dbcsr_kinds.F base/dbcsr_machine_internal.F base/dbcsr_machine.F base/dbcsr_base_hooks.F acc/dbcsr_acc_device.F; do
    amdflang $FFLAGS -c "$REF/src/$f" -o "$TMPF/$(basename $f .F).o"
  done
  amdflang $FFLAGS -c "$HERE/../tests/fortran/dbcsr_amd_host_check.F90" -o "$TMPF/host_check.o"
  amdflang -o "$OUT/fortran_host_check" "$TMPF"/*.o -L"$HERE/../dbcsr_amd" -ldbcsr_acc_amd -Wl,-rpath,'$ORIGIN/../../dbcsr_amd'
  rm -rf "$TMPF"
fi
echo "built $(ls $OUT | tr '\n' ' ')"
