#!/usr/bin/env python3
"""End-to-end check of the multi-rank driver on the HIP engine (development tool):
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/run_dist_check.py [backend]
Every rank runs dbcsr_amd.cannon.CannonMultiply on the GPU given by LOCAL_RANK modulo the
device count (so N ranks can share one GPU with backend=gloo), rank 0 compares the gathered C
with the CPU oracle's global multiply."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import torch.distributed as dist


def main():
    backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
    mode = sys.argv[2] if len(sys.argv) > 2 else "gather"
    transport = sys.argv[3] if len(sys.argv) > 3 else "torch"   # torch | native | auto (dbcsr_amd.cannon.CannonMultiply)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(backend, rank=rank, world_size=world)
    from dbcsr_amd import cannon
    from dbcsr_amd.multiply import MultiplyEngine
    M, N, K, sp = 23 * 60 + 16, 23 * 50 + 16, 23 * 70 + 16, (0.8, 0.8, 0.85)
    eps = None
    if mode.endswith("+filter"):   # on-the-fly filter + final block filter on several ranks (every rank decides as one rank would)
        mode, eps = mode[:-len("+filter")], 150.0
    if mode.endswith("+dist"):
        # distributed input: every rank holds an arbitrary share of the blocks of A, B and C, their data in HBM; make_images
        # (cannon.redistribute) packs, exchanges and sorts them on the device
        from oracle import oracle as O
        A, B, Cm = O.perf_case(M, N, K, *sp, [1, 23], [1, 23], [1, 23])

        def part(Mx, salt):
            rows = Mx.rows()
            mine = [b for b in range(Mx.nblks) if (b * 7 + salt * 3 + int(rows[b])) % world == rank]
            ne = [int(Mx.row_sizes[rows[b]]) * int(Mx.col_sizes[Mx.col_i[b]]) for b in mine]
            data = np.concatenate([Mx.data[Mx.blk_p[b]:Mx.blk_p[b] + n] for b, n in zip(mine, ne)]) if mine else np.zeros(0)
            z = np.zeros(0, np.int32)
            return cannon.DistBlocks(rows[mine] if mine else z, Mx.col_i[mine] if mine else z, torch.from_numpy(data).cuda())

        plan = cannon.CannonMultiply(dtype=torch.float64, engine=MultiplyEngine(), mode=mode.split("+")[0], transport=transport,
                                     distributed=((part(A, 1), part(B, 2), part(Cm, 3)), (A.row_sizes, A.col_sizes, B.col_sizes)))
    else:
        plan = cannon.CannonMultiply(M, N, K, sp, [1, 23], dtype=torch.float64, engine=MultiplyEngine(), mode=mode, transport=transport)
    for _ in range(2):
        Cout, counts = plan.multiply(0.5, 2.0, filter_eps=eps) if eps else plan.multiply(0.5, 2.0)
    torch.cuda.synchronize()
    if plan.mode in ("colpipe", "colpipe2d", "tilepipe") and not eps:   # the chunks of the second multiply went straight into their slices of one buffer
        assert plan.colpipe_copies == 1, plan.colpipe_copies
    parts = plan.gather_global(Cout)
    fl = torch.tensor([counts.flop], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(fl)
    ok = True
    if rank == 0:
        from oracle import oracle as O
        A, B, Cm = O.perf_case(M, N, K, *sp, [1, 23], [1, 23], [1, 23])
        ref, info = O.multiply("N", "N", 0.5, A, B, 2.0, Cm, filter_eps=eps or 0.0)
        if eps:
            full, finfo = O.multiply("N", "N", 0.5, A, B, 2.0, Cm)
            assert ref.nblks < full.nblks, "the filter case does not filter"
        got = {}
        for grow, gcol, blocks in parts:
            for r, c, blk in zip(grow, gcol, blocks):
                got[(int(r), int(c))] = blk
        rows = ref.rows()
        ok = len(got) == ref.nblks and int(fl.item()) == info["flop"]
        err = 0.0
        for b in range(ref.nblks):
            ne = int(ref.row_sizes[rows[b]]) * int(ref.col_sizes[ref.col_i[b]])
            exp = ref.data[ref.blk_p[b]:ref.blk_p[b] + ne]
            g = got.get((int(rows[b]), int(ref.col_i[b])))
            if g is None:
                ok = False
                break
            err = max(err, float(np.max(np.abs(g - exp) / np.maximum(np.abs(exp), 1e-300))))
        ok = ok and err <= 1e-10
        print("dist check mode=%s world=%d grid=%dx%d nvirt=%d blocks=%d transport=%s max_rel_err=%.2e flop_ok=%s -> %s" %
              (mode, world, plan.grid.nprows, plan.grid.npcols, plan.grid.nvirt, ref.nblks, plan.transport, err, int(fl.item()) == info["flop"],
               "OK" if ok else "FAIL"))
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
