#!/usr/bin/env python3
"""First contact with a multi-GPU node, stage 1: the panel exchange between REAL peers, checked byte for byte.
   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/comm_selfcheck.py [nccl|gloo] [native|torch]
Every rank sends every other rank a buffer whose contents are a function of (sender, receiver, position), in ONE grouped exchange per
size -- the pattern of the gather schedule: all links of a GPU busy at once (dbcsr_amd/cannon.py: _post_all; reference: the device-pointer
isend / irecv of multiply_cannon_g2g, src/mm/dbcsr_mm_cannon.F:2528-2557, 2655-2684) -- and checks what it received against the same function
evaluated locally; then the sizes allgather (make_images) against a host copy.  The largest size is timed: GB/s per link and direction with
all pairs active.  `native` = the C-ABI transport (include/dbcsr_amd_comm.h, RCCL grouped send / recv on a dedicated stream), `torch` =
torch.distributed point-to-point (the fall-back, and what the CPU dry run of tests/test_scale_first_contact.py uses with gloo)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist


def pattern(src, dst, n, device):
    i = torch.arange(n, dtype=torch.float64, device=device)
    return (i * 1.000001 + 1000.0 * src + 7.0 * dst + 0.25).to(torch.float64)


def main():
    backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
    transport = sys.argv[2] if len(sys.argv) > 2 else "native"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    gpu = backend == "nccl" or (torch.cuda.is_available() and os.environ.get("COMM_SELFCHECK_CPU") != "1")
    if gpu:
        ndev = torch.cuda.device_count()
        if backend == "nccl" and ndev < world:
            raise SystemExit("comm_selfcheck: %d ranks need %d devices (RCCL does not serve two ranks on one device); found %d" % (world, world, ndev))
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    dev = torch.device("cuda", torch.cuda.current_device()) if gpu else torch.device("cpu")
    dist.init_process_group(backend, rank=rank, world_size=world)
    comm = None
    if transport == "native":
        from dbcsr_amd.comm import NativeComm
        comm = NativeComm()
    sizes = [128, 131072, 8 * 1024 * 1024] if gpu else [128, 65536]   # doubles: 1 KiB, 1 MiB, 64 MiB per pair

    def exchange(sends, recvs):
        if comm is not None:
            ev = comm.exchange(sends, recvs)
            torch.cuda.current_stream().wait_event(ev)
            return
        ops = [dist.P2POp(dist.isend, t, p) for t, p in sends] + [dist.P2POp(dist.irecv, t, p) for t, p in recvs]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()

    def sync():
        if gpu:
            torch.cuda.synchronize()

    bad = 0
    rate = None
    for n in sizes:
        peers = [p for p in range(world) if p != rank]
        sends = [(pattern(rank, p, n, dev), p) for p in peers]
        recvs = [(torch.full((n,), -1.0, dtype=torch.float64, device=dev), p) for p in peers]
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        exchange(sends, recvs)
        sync()
        dt = time.perf_counter() - t0
        for buf, p in recvs:
            if not torch.equal(buf, pattern(p, rank, n, dev)):
                bad += 1
                sys.stderr.write("comm_selfcheck: rank %d got wrong data from %d at %d doubles\n" % (rank, p, n))
        if n == sizes[-1] and peers:
            for _ in range(2):   # timed repetitions of the largest exchange (connections are open now)
                sync()
                dist.barrier()
                t0 = time.perf_counter()
                exchange(sends, recvs)
                sync()
                dt = min(dt, time.perf_counter() - t0)
            rate = n * 8 / dt / 1e9   # GB/s per link and direction (every pair moves n doubles each way at once)
    # sizes allgather (make_images: every rank tells every rank how much it holds)
    mine = torch.tensor([rank * 3 + 1, rank * 5 + 2], dtype=torch.int64, device=dev)
    allv = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    if comm is not None:
        torch.cuda.current_stream().wait_event(comm.allgather_bytes(mine, allv))
    else:
        dist.all_gather_into_tensor(allv, mine) if backend == "nccl" else dist.all_gather(list(allv.view(world, 2)), mine)
    sync()
    want = torch.tensor([x for r in range(world) for x in (r * 3 + 1, r * 5 + 2)], dtype=torch.int64)
    if not torch.equal(allv.cpu(), want):
        bad += 1
        sys.stderr.write("comm_selfcheck: rank %d: allgather mismatch %s\n" % (rank, allv.cpu().tolist()))
    flag_dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([bad], dtype=torch.int64, device=flag_dev)
    dist.all_reduce(t)
    r = torch.tensor([rate or 0.0], dtype=torch.float64, device=flag_dev)
    dist.all_reduce(r, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"comm_selfcheck": "OK" if int(t.item()) == 0 else "FAILED", "ranks": world, "backend": backend, "transport": transport,
                          "rccl_ranks": comm.world if comm is not None else (world if backend == "nccl" else 0), "pairs_checked": world * (world - 1),
                          "largest_message_bytes": sizes[-1] * 8, "gb_per_s_per_link_all_pairs_active_min_over_ranks": round(float(r.item()), 3)}))
    if comm is not None:
        comm.close()
    dist.destroy_process_group()
    sys.exit(0 if int(t.item()) == 0 else 1)


if __name__ == "__main__":
    main()
