#!/usr/bin/env python3
"""Experiment: C += A*B in nchunk passes over k (submatrix limits first_k/last_k) against one pass, for workloads whose A
block-rows do not fit the L2 (config 5).  Prints ms per multiply for each variant."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402
from dbcsr_amd import randmat  # noqa: E402
from dbcsr_amd.matrix import DbcsrMatrix  # noqa: E402
from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "config5_131072_32x32_fill20_fp32"
chunks = [int(x) for x in sys.argv[2:]] or [1, 4, 8]
M, N, K, fill, mix, dt = bench.WORKLOADS[name]
dtype = torch.float64 if dt == "f64" else torch.float32
E = MultiplyEngine()
A, B, C0 = randmat.perf_matrices(M, N, K, (1 - fill,) * 3, mix, mix, mix, dtype=dtype, engine=E)
torch.cuda.synchronize()
ref = None
for nc in chunks:
    for rep in range(2):
        C = DbcsrMatrix(C0.row_blk_size, C0.col_blk_size, C0.row_p, C0.col_i, C0.blk_p, C0.data.clone(), "C")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if nc == 1:
            dbcsr_multiply("N", "N", 1.0, A, B, 1.0, C, engine=E)
        else:
            edges = [round(i * K / nc) for i in range(nc + 1)]
            for c in range(nc):
                dbcsr_multiply("N", "N", 1.0, A, B, 1.0, C, first_k=edges[c] + 1, last_k=edges[c + 1], engine=E)
        torch.cuda.synchronize()
        dt_ms = (time.perf_counter() - t0) * 1e3
    cs = E.checksum(C)
    if ref is None:
        ref = cs
    print("%s chunks=%d: %.1f ms per multiply, checksum rel diff %.2e" % (name, nc, dt_ms, abs(cs[0] - ref[0]) / abs(ref[0])), flush=True)
    del C
