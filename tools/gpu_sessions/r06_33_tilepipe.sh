#!/bin/bash
# round 6, GPU session 33: the tile pipeline on the 2-D grid (cannon.py mode "tilepipe": A's missing images in row chunks next to B's column chunks, step s multiplies the strips of C
# batch s completes) -- N ranks sharing the GPU against the oracle, then one rank's share of a step ALONE on the device, per strip (the inputs of tools/schedule_model.py)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
O=gpurun_out/r06_s33; mkdir -p $O
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
timeout 1500 python -m pytest tests/test_gpu_cannon_shared_gpu.py -q -m gpu -x -k "tilepipe" 2>&1 | tail -5 | tee $O/pytest.txt
for W in config2_32768_23x23_fill10_fp64 config4_131072_23x23_fill1_fp64; do
  for C in 4 6 8; do
    echo "# tilepipe (2-D grid), $C chunks per side" >> $O/budget.txt
    timeout 600 python tools/rank_step_budget.py --workload $W --ranks 4,8 --tilepipe $C 2>&1 | grep -v "$F" >> $O/budget.txt
  done
done
cat $O/budget.txt
