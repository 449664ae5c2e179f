#!/bin/bash
# what the reference's own performance driver measures on this box through the three host variants (H2O-like 23 x 23 blocks, 10 % fill)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/s12; mkdir -p $O
export MKL_THREADING_LAYER=SEQUENTIAL
python - <<'PY'
toks = [0, "F", "dbcsr_multiply", 16384, 16384, 16384, "0.9d0", "0.9d0", "0.9d0", "N", "N", "N", "N", "N", 3, "1.0d0", "0.0d0", "1.0d0", "0.0d0",
        0, 0, 0, 0, 0, 0, "F", 3, 1, 1, 1, 1, 23, 1, 23, 1, 23, "F", "0.1E-10", "0.0E+00", "0.0E+00"]
open("gpurun_out/s12/h2o16k.perf", "w").write("\n".join(str(t) for t in toks) + "\n")
PY
P=$PWD/$O/h2o16k.perf
for v in "host_cpu 32 0" "host_acc 8 0" "host_resident 8 1v"; do set -- $v
  ( cd /tmp && DBCSR_AMD_RESIDENT=$3 OMP_NUM_THREADS=$2 timeout 600 $OLDPWD/oracle/_ref/$1/dbcsr_perf $P > $OLDPWD/$O/$1.txt 2>&1 )
  echo "== $1 (OMP $2, resident $3)"; grep -E "dbcsr_amd_resident:|time  |perf total|flops total|matmuls total|checksum\(C_out\) " $O/$1.txt
done
