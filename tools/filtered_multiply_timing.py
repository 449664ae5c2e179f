#!/usr/bin/env python3
"""filtered multiply of sparse matrices (config 4's shape): unfiltered, filter that drops nothing, filter that drops about a third
of the blocks; candidate-driven against product-driven symbolic kernels"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from dbcsr_amd.multiply import MultiplyEngine
from dbcsr_amd.randmat import perf_matrices

SHAPE = os.environ.get("SHAPE", "config4")
SIZE, FILL, MIX, EPS = {"config4": (131072, 0.01, [1, 23], 140.0), "config3": (32768, 0.05, [1, 13, 1, 23, 1, 32], 300.0),
                        "config2": (32768, 0.10, [1, 23], 500.0)}[SHAPE]
for symbolic in (sys.argv[1:] or ["grid", "auto"]):
    if symbolic == "auto":
        os.environ.pop("DBCSR_AMD_MM_SYMBOLIC", None)
    else:
        os.environ["DBCSR_AMD_MM_SYMBOLIC"] = symbolic
    E = MultiplyEngine()
    A, B, Cm = perf_matrices(SIZE, SIZE, SIZE, (1 - FILL,) * 3, MIX, MIX, MIX, dtype=torch.float64, engine=E)
    for eps in (0.0, 1.0e-3, EPS):
        for _ in range(2):
            out, cnt = E.multiply_local(1.0, A, B, 1.0, Cm, filter_eps=eps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            out, cnt = E.multiply_local(1.0, A, B, 1.0, Cm, filter_eps=eps)
        torch.cuda.synchronize()
        print("symbolic=%s filter_eps=%g: %.2f ms per multiply, %d blocks before the final filter, %d after, %d products" %
              (symbolic, eps, (time.perf_counter() - t0) / 4 * 1e3, cnt.c_nblks, out.nblks, cnt.nproducts), flush=True)
        del out
    del A, B, Cm, E
