#!/usr/bin/env python3
"""Step budget of ONE rank of an N-GPU run, measured on one GPU (development tool).

For N in 1, 2, 4, 8 the block patterns of BASELINE config 2 (or --workload) are cut as dbcsr_amd.cannon does for an N-rank
grid, and the local multiply of rank 0 -- its full A row panel times its full B column panel into its C tile, the "gather"
schedule's one multiply per step -- is run alone on the GPU with synthetic values.  Reported per multiply: wall time of the call
sequence (symbolic + numeric as the driver issues them, then a device synchronisation), the block-product kernel's own time
(HIP events), and their difference = everything a rank spends per step outside the kernel (plan comparison and its one
synchronisation, kernel launches, Python, allocation of the result) -- the fixed cost that bounds strong scaling.
   python tools/rank_step_budget.py [--workload NAME] [--plan 0|1]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--workload", default="config2_32768_23x23_fill10_fp64")
    p.add_argument("--ranks", default="1,2,4,8")
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--colpipe", type=int, default=0, help="N x 1 grid and the colpipe schedule with this many column chunks (its whole compute path, "
                                                          "exchange left out): kernel_ms is then the SUM over the chunk multiplies")
    p.add_argument("--colpipe2d", type=int, default=0, help="the same column-chunk pipeline on the default 2-D grid (round 6) with this many chunks")
    p.add_argument("--tilepipe", type=int, default=0, help="the row-chunk x column-chunk tile pipeline on the 2-D grid with this many chunks per side: "
                                                           "kernel_ms is the SUM over the strips, the per-strip times are printed too")
    a = p.parse_args()
    if a.colpipe2d:
        a.colpipe = a.colpipe2d
    if a.tilepipe:
        a.colpipe = a.tilepipe
    import bench
    from dbcsr_amd import cannon
    from dbcsr_amd.multiply import MultiplyEngine
    M, N, K, fill, mix, dt = bench.WORKLOADS[a.workload]
    dtype = torch.float64 if dt == "f64" else torch.float32
    print("# workload %s; plan reuse %s" % (a.workload, os.environ.get("DBCSR_AMD_MM_PLAN", "1")))
    print("# ranks grid  C_blocks  products   GFLOP   wall_ms  kernel_ms  fill_ms  non_kernel_ms  kernel")
    for n in [int(x) for x in a.ranks.split(",")]:
        eng = MultiplyEngine()
        g = cannon.Grid(n, 0, nprows=n, npcols=1) if (a.colpipe and not a.colpipe2d and not a.tilepipe) else cannon.Grid(n, 0)
        plan = cannon.CannonMultiply(M, N, K, (1 - fill,) * 3, mix, dtype=dtype, engine=eng, grid=g,
                                     mode="tilepipe" if a.tilepipe else "colpipe2d" if a.colpipe2d else ("colpipe" if a.colpipe else "gather"), col_chunks=max(1, a.colpipe))
        # images owned by other ranks: synthetic values in place (what would have arrived over xGMI)
        for buf in (plan._a_all, plan._b_all):
            buf.uniform_(0.0, 1.0)
        torch.cuda.synchronize()

        def step():
            row_p, counts = eng.symbolic(plan.A_panel, plan.B_panel, plan.C_in, retain_sparsity=False)
            out = eng.numeric_after_symbolic(1.0, plan.A_panel, plan.B_panel, 1.0, plan.C_in, row_p, counts, dtype)
            return out, counts

        kernel_time = lambda: eng.last_timing()
        if a.colpipe:
            plan._exchange = lambda sends, recvs: ([], [])   # nothing travels here: the panels are in place

            def step():
                return plan.multiply(1.0, 1.0)

            per_strip = []

            def kernel_time():
                engines = [plan._engine(("tile", i)) for i in range(len(plan._tiles))] if a.tilepipe else \
                    [plan._engine(("col", q)) for q in range(len(plan._cbounds) - 1)]
                t = [e.last_timing() for e in engines]
                per_strip[:] = [round(x[1], 3) for x in t]
                return sum(x[0] for x in t), sum(x[1] for x in t)

        for _ in range(3):
            out, counts = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out, counts = step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.steps * 1e3
        ks, fs = [], []
        for _ in range(3):
            out, counts = step()
            f, k = kernel_time()
            ks.append(k)
            fs.append(f)
        km, fm = sum(ks) / len(ks), sum(fs) / len(fs)
        print("%7d %dx%d %9d %9d %8.1f %9.3f %9.3f %8.3f %12.3f   %s" % (n, g.nprows, g.npcols, counts.c_nblks, counts.nproducts, counts.flop / 1e9, wall, km,
                                                                       fm, wall - km, eng.last_kernel()))
        if a.tilepipe:
            print("#   strips (step, rows, columns): %s" % ", ".join("(%d, %d-%d, %d-%d)" % t for t in plan._tiles))
            print("#   kernel ms per strip: %s" % per_strip)
        del plan, eng, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
