#!/bin/bash
# round 4, GPU session 8: k passes on index views with one engine per pass (config 5), and config 2 through the resident loop on 1 / 2 / 4 ranks
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r04_s08; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multiply.py -k "k_chunked" tests/test_gpu_fullsize.py -x -q 2>&1 | tail -6 | tee $O/pytest.txt
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/config5.txt
import sys, json
sys.argv = ["bench.py"]
import bench
for w in ("config5_131072_32x32_fill20_fp32", "fp32_16384_32x32_fill20"):
    r = bench.run_other_config(w, steps=3, warmup=1)
    print(json.dumps(r))
PY
export MKL_THREADING_LAYER=SEQUENTIAL OMP_NUM_THREADS=4 DBCSR_AMD_RESIDENT=0
MPIEXEC=$(which mpiexec || echo /opt/conda/bin/mpiexec)
for n in 1 2 4; do
  echo "== config 2 (32768, 23 x 23, 10 %), resident loop, $n rank(s) sharing the GPU" | tee -a $O/resident_loop_config2.txt
  timeout 900 $MPIEXEC -n $n oracle/_ref/host_resident_mpi/dbcsr_resident_loop 32768 0.9 23 8 0 2>&1 | grep "resident_loop\|rror" | tee -a $O/resident_loop_config2.txt
done
