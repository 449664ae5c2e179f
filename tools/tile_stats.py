#!/usr/bin/env python3
"""Config 2 with the tile kernel: kernel time, the kernel's own time breakdown (DBCSR_AMD_MM_TILE_KNOBS=32) and protocol statistics.
Environment: DBCSR_AMD_MM_TILE=2, DBCSR_AMD_MM_TILE_SHAPE / _WINDOW / _KNOBS, DBCSR_AMD_MM_TILE_VERBOSE=1."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dbcsr_amd import randmat
from dbcsr_amd.multiply import MultiplyEngine

eng = MultiplyEngine()
A, B, C = randmat.perf_matrices(32768, 32768, 32768, (0.9, 0.9, 0.9), [1, 23], [1, 23], [1, 23], dtype=torch.float64, engine=eng)
for _ in range(3):
    out, counts = eng.multiply_local(1.0, A, B, 1.0, C)
torch.cuda.synchronize()
print("shape", os.environ.get("DBCSR_AMD_MM_TILE_SHAPE"), "window", os.environ.get("DBCSR_AMD_MM_TILE_WINDOW"), "knobs",
      os.environ.get("DBCSR_AMD_MM_TILE_KNOBS"), "kernel ms %.3f" % eng.last_timing()[1], eng.last_kernel(), eng.tile_stats())
