"""N > 1 path on CPU: the Cannon driver (distribution, skewed schedule, owner-direct panel
exchange, in-place tick accumulation) under the gloo backend with world sizes 2, 4, 6 and 8
(2x1, 2x2, the non-square 3x2 grid with nvirt = 6 and BASELINE's 4x2 grid with nvirt = 4), checked against the oracle's global
multiply.  The local arithmetic is the oracle here (tests/cpu_backend.py); on a GPU node the
same driver runs on the HIP engine with the nccl (= RCCL) backend."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASE = dict(M=23 * 9 + 16, N=13 * 14 + 5, K=17 * 8 + 3, sp=(0.6, 0.65, 0.8), mix=[1, 23], mix_n=[1, 13], mix_k=[2, 17, 1, 5])


def _worker(rank, world, port, alpha, beta, q, mode, retain=False, eps=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dbcsr_amd import cannon
        from tests.cpu_backend import OracleBackend
        if mode.endswith("+dist"):  # distributed input: every rank holds an arbitrary third / quarter ... of the blocks
            A, B, Cm = O.perf_case(CASE["M"], CASE["N"], CASE["K"], *CASE["sp"], CASE["mix"], CASE["mix_n"], CASE["mix_k"])

            def part(M, salt):
                rows = M.rows()
                mine = [b for b in range(M.nblks) if (b * 7 + salt * 3 + int(rows[b])) % world == rank]
                ne = [int(M.row_sizes[rows[b]]) * int(M.col_sizes[M.col_i[b]]) for b in mine]
                data = np.concatenate([M.data[M.blk_p[b]:M.blk_p[b] + n] for b, n in zip(mine, ne)]) if mine else np.zeros(0)
                return cannon.DistBlocks(rows[mine] if mine else np.zeros(0, np.int32), M.col_i[mine] if mine else np.zeros(0, np.int32), data)

            plan = cannon.CannonMultiply(dtype=torch.float64, engine=OracleBackend(), device=torch.device("cpu"), mode=mode.split("+")[0],
                                         distributed=((part(A, 1), part(B, 2), part(Cm, 3)), (A.row_sizes, A.col_sizes, B.col_sizes)))
        elif mode.endswith("+given"):  # replicated global host matrices instead of the built-in generator
            A, B, Cm = O.perf_case(CASE["M"], CASE["N"], CASE["K"], *CASE["sp"], CASE["mix"], CASE["mix_n"], CASE["mix_k"])
            plan = cannon.CannonMultiply(dtype=torch.float64, engine=OracleBackend(), device=torch.device("cpu"),
                                         mode=mode.split("+")[0], matrices=(A, B, Cm))
        else:
            plan = cannon.CannonMultiply(CASE["M"], CASE["N"], CASE["K"], CASE["sp"], CASE["mix"], dtype=torch.float64,
                                         engine=OracleBackend(), device=torch.device("cpu"), mix_n=CASE["mix_n"], mix_k=CASE["mix_k"],
                                         mode=mode)
        Cout, counts = plan.multiply(alpha, beta, retain_sparsity=retain, filter_eps=eps) if (retain or eps) else plan.multiply(alpha, beta)
        parts = plan.gather_global(Cout)
        fl = torch.tensor([counts.flop], dtype=torch.int64)
        dist.all_reduce(fl)
        if rank == 0:
            q.put((parts, int(fl.item()), plan.grid.nprows, plan.grid.npcols, plan.grid.nvirt))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "gather"), (4, "gather+given"), (6, "gather"), (4, "ticks"), (6, "ticks+given"),
                                        (4, "ticks+dist"), (6, "gather+dist"), (2, "ticks+dist"), (8, "ticks"), (8, "gather+dist"),
                                        (2, "colpipe"), (3, "colpipe+given"), (4, "colpipe+dist"), (8, "colpipe"),
                                        # round 6: the column-chunk pipeline on the 2-D grid (2 x 2, 3 x 2, 4 x 2)
                                        (4, "colpipe2d"), (6, "colpipe2d+given"), (8, "colpipe2d"), (4, "colpipe2d+dist"),
                                        # ... and the row-chunk x column-chunk tile pipeline (A's images chunked too)
                                        (4, "tilepipe"), (6, "tilepipe+given"), (8, "tilepipe"), (4, "tilepipe+dist"), (2, "tilepipe")])
def test_cannon_matches_global_oracle(world, mode):
    _run_and_compare(world, mode)


@pytest.mark.parametrize("world,mode,retain,eps", [(4, "gather", False, 60.0), (6, "ticks", False, 80.0), (4, "ticks+dist", True, None),
                                                  (2, "gather", True, 60.0), (3, "colpipe", False, 60.0), (2, "colpipe+dist", True, None),
                                                  (4, "colpipe2d", False, 60.0), (4, "tilepipe", False, 60.0), (6, "tilepipe", True, None)])
def test_cannon_filter_and_retain_match_global_oracle(world, mode, retain, eps):
    """filter_eps / retain_sparsity on several ranks: every rank takes the decisions one rank would (row counts of the WHOLE block
    row enter the on-the-fly filter, dbcsr_mm_cannon.F:1040-1113), block structure and values equal the single-rank oracle's."""
    _run_and_compare(world, mode, retain, eps)


def _run_and_compare(world, mode, retain=False, eps=None):
    alpha, beta = 0.75, -1.25
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, alpha, beta, q, mode, retain, eps)) for r in range(world)]
    for p in procs:
        p.start()
    parts, flop, pr, pc, nvirt = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert pr * pc == world and nvirt == pr * pc // np.gcd(pr, pc)
    # global reference: same generator, same seeds, one rank
    A, B, Cm = O.perf_case(CASE["M"], CASE["N"], CASE["K"], *CASE["sp"], CASE["mix"], CASE["mix_n"], CASE["mix_k"])
    ref, info = O.multiply("N", "N", alpha, A, B, beta, Cm, retain_sparsity=retain, filter_eps=eps or 0.0)
    assert flop == info["flop"]  # every block product executed exactly once across ranks and ticks
    got = {}
    for grow, gcol, blocks in parts:
        for r, c, blk in zip(grow, gcol, blocks):
            assert (int(r), int(c)) not in got  # C is partitioned, no block on two ranks
            got[(int(r), int(c))] = blk
    rows = ref.rows()
    assert len(got) == ref.nblks  # identical block structure (as a set of global coordinates)
    for b in range(ref.nblks):
        key = (int(rows[b]), int(ref.col_i[b]))
        ne = int(ref.row_sizes[rows[b]]) * int(ref.col_sizes[ref.col_i[b]])
        exp = ref.data[ref.blk_p[b]:ref.blk_p[b] + ne]
        assert np.allclose(got[key], exp, rtol=1e-12, atol=1e-12), key


def test_grid_and_schedule_properties():
    from dbcsr_amd import cannon
    assert [cannon.dims_create(n) for n in (1, 2, 4, 8)] == [(1, 1), (2, 1), (2, 2), (4, 2)]  # SURVEY 8e grids
    for n in (1, 2, 3, 4, 6, 8, 12):
        pr, pc = cannon.dims_create(n)
        g0 = cannon.Grid(n, 0)
        for t in range(g0.nvirt):
            for r in range(pr):
                # in every tick the ranks of a process row use distinct A images, each owned inside that row
                vs = [g0.v_at(r, c, t) for c in range(pc)]
                assert len(set(vs)) == pc
                assert all(g0.a_owner(r, v) // pc == r for v in vs)
            for c in range(pc):
                vs = [g0.v_at(r, c, t) for r in range(pr)]
                assert len(set(vs)) == pr
                assert all(g0.b_owner(v, c) % pc == c for v in vs)
    assert list(cannon.dist_bin([5, 5, 5, 3, 1], 2)) == [0, 1, 0, 1, 1]


class _FakeComm:
    """Same call interface as dbcsr_amd.comm.NativeComm (allgather_bytes / exchange returning something to synchronize on),
    carried by gloo: checks the call sequence redistribute() makes on the C-ABI transport."""

    class _Done:
        def synchronize(self):
            pass

    def exchange(self, sends, recvs):
        ops = [dist.P2POp(dist.isend, t, d) for t, d in sends] + [dist.P2POp(dist.irecv, t, d) for t, d in recvs]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return self._Done()

    def allgather_bytes(self, t_send, t_recv):
        dist.all_gather([t_recv[r] for r in range(t_recv.shape[0])], t_send)
        return self._Done()


def _redist_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dbcsr_amd import cannon
        rng = np.random.default_rng(11)
        rs = rng.integers(1, 9, size=17).astype(np.int32)
        cs = rng.integers(1, 9, size=13).astype(np.int32)
        present = np.argwhere(rng.random((17, 13)) < 0.5)
        rows_g, cols_g = present[:, 0].astype(np.int32), present[:, 1].astype(np.int32)
        vals = {(int(r), int(c)): rng.standard_normal(int(rs[r]) * int(cs[c])) for r, c in zip(rows_g, cols_g)}
        holder = (rows_g * 5 + cols_g * 3) % world           # where a block starts
        dest_of = lambda rr, cc: (rr.astype(np.int64) + 2 * cc) % world   # where it has to go
        mine = np.nonzero(holder == rank)[0][::-1]           # in no particular order
        loc = cannon.DistBlocks(rows_g[mine], cols_g[mine],
                                np.concatenate([vals[(int(rows_g[b]), int(cols_g[b]))] for b in mine]) if len(mine) else np.zeros(0))
        res = []
        for comm in (None, _FakeComm()):
            (rows, cols, off, data), everyone = cannon.redistribute(loc, dest_of, rs, cs, torch.float64, comm=comm)
            ok = all(int(dest_of(np.asarray([r]), np.asarray([c]))[0]) == rank for r, c in zip(rows, cols))
            ok = ok and np.array_equal(np.lexsort((cols, rows)), np.arange(len(rows)))
            ok = ok and all(np.array_equal(data[off[b]:off[b + 1]].numpy(), vals[(int(rows[b]), int(cols[b]))]) for b in range(len(rows)))
            total = sum(len(e[0]) for e in everyone)
            ok = ok and total == len(rows_g) and np.array_equal(everyone[rank][0], rows)
            res.append(bool(ok))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_redistribute_torch_and_c_abi_call_sequence():
    """make_images for one matrix on 3 ranks: every block arrives at the owner of its image, sorted, bit-identical; the same
    through the call interface of the C-ABI exchange (allgather of sizes + one grouped exchange)."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_redist_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert all(res == [True, True] for _, res in got), got
