#!/bin/bash
# BASELINE config 2 (32768^2, 23 x 23 blocks, 10 % fill, fp64) through the reference's OWN performance driver: unchanged CPU build and the
# patched host (dbcsr_multiply -> device-resident engine; A, B, C live in host memory and cross PCIe in every call)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/refdriver_c2; mkdir -p $O
export MKL_THREADING_LAYER=SEQUENTIAL
python - <<'PY'
toks = [0, "F", "dbcsr_multiply", 32768, 32768, 32768, "0.9d0", "0.9d0", "0.9d0", "N", "N", "N", "N", "N", 3, "1.0d0", "0.0d0", "1.0d0", "0.0d0",
        0, 0, 0, 0, 0, 0, "F", 3, 1, 1, 1, 1, 23, 1, 23, 1, 23, "F", "0.1E-10", "0.0E+00", "0.0E+00"]
open("gpurun_out/refdriver_c2/config2.perf", "w").write("\n".join(str(t) for t in toks) + "\n")
PY
P=$PWD/$O/config2.perf
for v in "host_resident 8 1v" "host_cpu 32 0"; do set -- $v
  [ "${ONLY_RESIDENT:-0}" = 1 ] && [ $1 != host_resident ] && continue
  ( cd /tmp && DBCSR_AMD_RESIDENT=$3 OMP_NUM_THREADS=$2 timeout 900 $OLDPWD/oracle/_ref/$1/dbcsr_perf $P > $OLDPWD/$O/$1.txt 2>&1 )
  echo "== $1 (OMP $2, resident $3)"; grep -E "dbcsr_amd_resident:|time  |perf total|flops total|matmuls total|checksum\(C_out\) " $O/$1.txt
done
