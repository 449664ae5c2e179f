"""Multi-GPU block-sparse multiply: the 2-D distributed algorithm of
``multiply_cannon`` (reference src/mm/dbcsr_mm_cannon.F:839-1771, image
distributions of src/dist/dbcsr_dist_methods.F:423-454), one process per GPU.

What is kept from the reference: the 2-D block distribution of C over an
``nprows x npcols`` grid (block rows/columns binned by size, dbcsr_dist_bin,
src/dist/dbcsr_dist_operations.F:708-745), C stationary, A and B cut into
``nvirt = lcm(nprows, npcols)`` virtual k-images, the skewed schedule (rank (r, c)
multiplies image ``v = (r + c + t) mod nvirt`` at tick t) so that every tick each
rank needs exactly one A image from its process row and one B image from its
process column, and double buffering of the next tick's panels against the
current tick's local multiply.

What is MI355X-native: panels are fetched DIRECTLY FROM THEIR OWNER with RCCL
send/recv (every GPU pair of a node has its own xGMI link, so a ring that forwards
panels hop by hop would only add hops), on the communication stream of
torch.distributed while the local multiply runs on the compute stream; index
metadata never travels (every rank derives all image indices from the replicated
block pattern), only block data does; C's structure for all ticks is computed
once on the GPU and the ticks accumulate in place.

Three schedules are offered:
  * ``mode="gather"`` (default): every rank posts ONE batch that fetches all the images it does not own
    -- all peers, hence all xGMI links, at the same time -- while the GPU already runs the symbolic phase
    (which needs index metadata only); then one device-resident multiply over the full row/column panels.
    With 288 GB per GPU the panels always fit, C is written exactly once, and there is one symbolic pass
    instead of one per tick.
  * ``mode="ticks"``: the reference's tick-by-tick pipeline (one A and one B image per tick, next tick's
    images in flight during the current tick's multiply, in-place accumulation into C).
  * ``mode="colpipe"`` (chosen at construction; an ``N x 1`` grid): A's block rows and C's stay where they are, B --
    one k-image per rank, stored column chunk by column chunk -- travels from every owner to everybody in column
    chunks, and chunk q of C is multiplied as soon as chunk q of B is complete while chunk q + 1 is on the links.
    Every C block is written once (the chunks land in slices of one buffer under a merged index), the exposed
    transfer is one chunk of one image per link, and odd chunks are multiplied on a second stream.
  * ``mode="colpipe2d"`` (round 6; chosen at construction): the same pipeline on the 2-D grid of the other schedules
    (``MPI_Dims_create``: 4 x 2 for eight ranks -- the redistribution the reference's Cannon loop uses,
    dbcsr_mm_cannon.F:1376-1463, 1497-1586).  The A images a rank misses (its process row's) come with the first batch;
    the B images of its process column travel in column chunks of its OWN block columns and chunk q of its C tile is
    multiplied when chunk q of every image has landed.  Per link the same bytes as on the N x 1 grid (one image per
    peer), in total fewer (a rank needs its row panel of A and its column panel of B, not all of B).
  * ``mode="tilepipe"`` (round 6; chosen at construction; the 2-D grid): colpipe2d leaves the A images of the process row exposed
    before the first multiply -- every column chunk needs the whole row panel of A.  Here A's missing images travel in ROW chunks
    of the rank's own block rows next to B's column chunks (batch s = row chunk s of A + column chunk s of B, A and B on different
    links), and step s multiplies the tiles of C that have become computable with batch s: the row strip
    (row chunk s) x (column chunks 0 .. s) and the column strip (row chunks 0 .. s - 1) x (column chunk s), two launches on two
    streams (disjoint C blocks, every C block written once).  Exposed: one chunk of one image per link; the work that can run
    grows with the square of what has landed (1, 3, 5, ... of S^2 tiles).

The local engine is duck-typed (``symbolic``, ``init_c``, ``accumulate``,
``fill_random_dist`` of dbcsr_amd.multiply.MultiplyEngine) so that the
distribution / schedule / communication logic can be exercised on CPU with the
``gloo`` backend in tests.
"""
import contextlib
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import randmat
from .matrix import DbcsrMatrix


class _EventWork:
    """wait() makes the current (compute) stream wait for a transfer posted on the communication stream"""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def dims_create(n):
    """MPI_Dims_create-like 2-D factorisation, non-increasing (8 -> 4 x 2), as the reference's
    default grid (src/mpi/dbcsr_mpiwrap.F:1107)."""
    d = int(math.isqrt(n))
    while n % d:
        d -= 1
    return n // d, d


def dist_bin(sizes, nbins):
    """Greedy least-loaded binning of elements in order (dbcsr_dist_bin); ties go to the lowest bin."""
    load = np.zeros(nbins, np.int64)
    out = np.empty(len(sizes), np.int32)
    for i, s in enumerate(sizes):
        b = int(np.argmin(load))
        out[i] = b
        load[b] += int(s)
    return out


class Grid:
    def __init__(self, world, rank, nprows=None, npcols=None, nvirt=None):
        if nprows is None:
            nprows, npcols = dims_create(world)
        assert nprows * npcols == world
        self.world, self.rank, self.nprows, self.npcols = world, rank, nprows, npcols
        self.myprow, self.mypcol = divmod(rank, npcols)
        lcm = nprows * npcols // math.gcd(nprows, npcols)
        self.nvirt = lcm if nvirt is None else nvirt  # any multiple of lcm works (more, thinner k-images)
        assert self.nvirt % lcm == 0

    def rank_of(self, prow, pcol):
        return prow * self.npcols + pcol

    def v_at(self, prow, pcol, tick):
        return (prow + pcol + tick) % self.nvirt

    def a_owner(self, prow, v):
        return self.rank_of(prow, v % self.npcols)

    def b_owner(self, v, pcol):
        return self.rank_of(v % self.nprows, pcol)


class Partition:
    """Block rows/columns/k-blocks -> process rows/columns/virtual images, with local numbering."""

    def __init__(self, row_sizes, k_sizes, col_sizes, grid):
        self.row_sizes, self.k_sizes, self.col_sizes = (np.asarray(x, np.int32) for x in (row_sizes, k_sizes, col_sizes))
        self.row_dist = dist_bin(self.row_sizes, grid.nprows)
        self.col_dist = dist_bin(self.col_sizes, grid.npcols)
        self.k_dist = dist_bin(self.k_sizes, grid.nvirt)
        self.rows_of = [np.nonzero(self.row_dist == r)[0].astype(np.int32) for r in range(grid.nprows)]
        self.cols_of = [np.nonzero(self.col_dist == c)[0].astype(np.int32) for c in range(grid.npcols)]
        self.ks_of = [np.nonzero(self.k_dist == v)[0].astype(np.int32) for v in range(grid.nvirt)]
        self.row_local = self._local(self.row_dist, self.rows_of)
        self.col_local = self._local(self.col_dist, self.cols_of)
        self.k_local = self._local(self.k_dist, self.ks_of)

    @staticmethod
    def _local(dist_arr, groups):
        loc = np.empty(len(dist_arr), np.int32)
        for g in groups:
            loc[g] = np.arange(len(g), dtype=np.int32)
        return loc


def _sub_index(rows, cols, keep, row_local, col_local, nrows_local, row_sizes, col_sizes):
    """BCSR index (row_p, col_i, blk_p, nze) of the kept blocks in local numbering.  The global
    pattern is sorted by (row, col) and local ids are monotone in global ids, so order is kept."""
    r, c = row_local[rows[keep]], col_local[cols[keep]]
    nze = row_sizes[rows[keep]].astype(np.int64) * col_sizes[cols[keep]].astype(np.int64)
    blk_p = np.zeros(len(r), np.int64)
    if len(r):
        blk_p[1:] = np.cumsum(nze)[:-1]
    row_p = np.zeros(nrows_local + 1, np.int64)
    np.add.at(row_p, r.astype(np.int64) + 1, 1)
    return np.cumsum(row_p).astype(np.int32), c.astype(np.int32), blk_p, int(nze.sum())


class DistBlocks:
    """The part of a distributed matrix that ONE rank holds before the multiply: any subset of the blocks, in any order,
    addressed by GLOBAL block coordinates (the counterpart of a rank-local dbcsr_type: src/core/dbcsr_types.F:362-461).
    rows / cols: int32 arrays (host), data: the blocks concatenated (column-major each) in the same order -- a numpy array
    or a torch tensor; a tensor in HBM stays there through the redistribution."""

    def __init__(self, rows, cols, data):
        self.rows, self.cols = np.ascontiguousarray(rows, np.int32), np.ascontiguousarray(cols, np.int32)
        self.data = data if isinstance(data, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(data))


_GATHER_CHUNK = 1 << 26  # elements per gather pass (bounds the int64 index vector to 512 MB)


def gather_blocks(src, starts, lens):
    """out = the blocks src[starts[b] : starts[b] + lens[b]] back to back, on src's device (starts / lens: host int64)."""
    starts, lens = np.asarray(starts, np.int64), np.asarray(lens, np.int64)
    total = int(lens.sum())
    out = torch.empty(total, dtype=src.dtype, device=src.device)
    if total == 0:
        return out
    dst = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    b0 = 0
    while b0 < len(lens):
        b1 = int(np.searchsorted(dst, dst[b0] + _GATHER_CHUNK, side="right"))
        b1 = max(b0 + 1, min(b1 - 1, len(lens)))
        n = int(dst[b1] - dst[b0])
        if n:
            shift = torch.as_tensor(starts[b0:b1] - dst[b0:b1], device=src.device)
            idx = torch.repeat_interleave(shift, torch.as_tensor(lens[b0:b1], device=src.device), output_size=n)
            idx += torch.arange(int(dst[b0]), int(dst[b1]), device=src.device)
            out[int(dst[b0]):int(dst[b1])] = src[idx]
        b0 = b1
    return out


def _wire_device(like):
    """Where point-to-point buffers must live: HBM under the nccl (= RCCL) backend, host memory under gloo."""
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return like.device if like.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def redistribute(loc, dest_of, row_sizes, col_sizes, dtype, comm=None):
    """make_images for one matrix (reference src/mm/dbcsr_mm_cannon.F:292-750): every block travels from the rank that
    holds it to the rank that owns its target image, as an (index, data) pair -- sizes first (one allgather, :532),
    then int32 block coordinates and the block data (:674-678, :1036).  Collective over the default process group.
    The int32 index is handled on the host (as the reference does); the block data is packed per destination, sent and
    sorted into (row, col) order ON THE DEVICE IT LIVES ON (gather_blocks) -- device buffers go straight into RCCL under
    the nccl backend, under gloo they are staged through host memory for the wire only; with ``comm`` (a
    dbcsr_amd.comm.NativeComm) sizes, index and data travel through the C-ABI exchange of include/dbcsr_amd_comm.h instead.
    Returns (rows, cols, offsets, data) of the blocks this rank now owns, sorted by (row, col) -- data a tensor on the
    input's device -- and the coordinates of EVERY block of the matrix with its owner (the index metadata the other
    ranks need to address those images)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    src = loc.data.to(dtype)
    dev = src.device
    nblk = len(loc.rows)
    nze = row_sizes[loc.rows].astype(np.int64) * col_sizes[loc.cols].astype(np.int64) if nblk else np.zeros(0, np.int64)
    off = np.concatenate([[0], np.cumsum(nze)]).astype(np.int64)
    dest = np.asarray(dest_of(loc.rows, loc.cols), np.int64) if nblk else np.zeros(0, np.int64)
    perm = np.argsort(dest, kind="stable")
    nb_to = np.bincount(dest, minlength=world).astype(np.int64)
    ne_to = np.bincount(dest, weights=nze, minlength=world).astype(np.int64) if nblk else np.zeros(world, np.int64)
    b_lo = np.concatenate([[0], np.cumsum(nb_to)]).astype(np.int64)
    e_lo = np.concatenate([[0], np.cumsum(ne_to)]).astype(np.int64)
    packed = gather_blocks(src, off[perm], nze[perm])            # grouped by destination rank
    idx_out = np.stack([loc.rows[perm], loc.cols[perm]]).astype(np.int32) if nblk else np.zeros((2, 0), np.int32)
    if world > 1:
        wire = src.device if comm is not None else _wire_device(src)
        # sizes: what every rank will send to every rank (blocks, elements)
        mine = torch.as_tensor(np.stack([nb_to, ne_to], axis=1).astype(np.int64)).to(wire)
        if comm is not None:   # C-ABI transport (include/dbcsr_amd_comm.h): allgather + ONE grouped exchange of device buffers
            gathered = torch.empty((world,) + tuple(mine.shape), dtype=torch.int64, device=wire)
            comm.allgather_bytes(mine, gathered).synchronize()
            allsz = gathered.cpu().numpy()
        else:
            allsz_t = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allsz_t, mine)
            allsz = np.stack([a.cpu().numpy() for a in allsz_t])
        nb_from = allsz[:, rank, 0].astype(np.int64)
        ne_from = allsz[:, rank, 1].astype(np.int64)
        rb_lo = np.concatenate([[0], np.cumsum(nb_from)]).astype(np.int64)
        re_lo = np.concatenate([[0], np.cumsum(ne_from)]).astype(np.int64)
        recv_dat = torch.empty(int(re_lo[-1]), dtype=dtype, device=dev)
        recv_idx = np.zeros((2, int(rb_lo[-1])), np.int32)
        sends, recvs, idx_bufs, dat_bufs = [], [], {}, {}
        for d in range(world):
            if d == rank:
                continue
            if nb_to[d]:
                ti = torch.from_numpy(np.ascontiguousarray(idx_out[:, b_lo[d]:b_lo[d + 1]])).to(wire)
                td = packed[int(e_lo[d]):int(e_lo[d + 1])].to(wire)
                sends += [(ti, d), (td, d)]
            if nb_from[d]:
                idx_bufs[d] = torch.empty((2, int(nb_from[d])), dtype=torch.int32, device=wire)
                seg = recv_dat[int(re_lo[d]):int(re_lo[d + 1])]
                dat_bufs[d] = seg if seg.device == wire else torch.empty(int(ne_from[d]), dtype=dtype, device=wire)
                recvs += [(idx_bufs[d], d), (dat_bufs[d], d)]
        if comm is not None:
            comm.exchange(sends, recvs).synchronize()
        else:
            ops = [dist.P2POp(dist.isend, t, d) for t, d in sends] + [dist.P2POp(dist.irecv, t, d) for t, d in recvs]
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            if wire.type == "cuda":
                torch.cuda.synchronize()
        for d in idx_bufs:
            recv_idx[:, rb_lo[d]:rb_lo[d + 1]] = idx_bufs[d].cpu().numpy()
            seg = recv_dat[int(re_lo[d]):int(re_lo[d + 1])]
            if dat_bufs[d].data_ptr() != seg.data_ptr():
                seg.copy_(dat_bufs[d])
        # my own share never touches the wire
        recv_idx[:, rb_lo[rank]:rb_lo[rank + 1]] = idx_out[:, b_lo[rank]:b_lo[rank + 1]]
        recv_dat[int(re_lo[rank]):int(re_lo[rank + 1])] = packed[int(e_lo[rank]):int(e_lo[rank + 1])]
        rows, cols, data = recv_idx[0].copy(), recv_idx[1].copy(), recv_dat
    else:
        rows, cols, data = idx_out[0].copy(), idx_out[1].copy(), packed
    n_in = row_sizes[rows].astype(np.int64) * col_sizes[cols].astype(np.int64) if len(rows) else np.zeros(0, np.int64)
    o_in = np.concatenate([[0], np.cumsum(n_in)]).astype(np.int64)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    sdata = gather_blocks(data, o_in[order], n_in[order])
    soff = np.concatenate([[0], np.cumsum(n_in[order])]).astype(np.int64)
    # index metadata of the whole matrix: every rank publishes the coordinates of the blocks it now owns
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, (rows, cols))
    else:
        everyone = [(rows, cols)]
    return (rows, cols, soff, sdata), everyone


class CannonMultiply:
    """Distributed C <- beta*C + alpha*A*B.

    Inputs are either synthetic matrices of the reference's generator (M, N, K, sparsities, block-size mix:
    every rank builds the same global block patterns on the host, O(nblks), and generates the values of its own
    part directly in HBM), or -- ``matrices=(A, B, C)`` -- global host matrices replicated on every rank (objects
    with row_sizes, col_sizes, row_p, col_i, blk_p, data), from which each rank cuts its part: the counterpart of
    the reference's ``make_m2s``/``make_images`` (dbcsr_mm_cannon.F:146-258, 292-750) for replicated input.
    A rank materialises its C tile, its A images (process row r, images v = c mod npcols) and its B images
    (process column c, images v = r mod nprows)."""

    def __init__(self, M=0, N=0, K=0, sparsities=(0, 0, 0), mix=(1, 1), dtype=torch.float64, engine=None, device=None, grid=None,
                 mix_n=None, mix_k=None, mode="gather", local_first=True, matrices=None, transport="torch", distributed=None,
                 col_chunks=8, share_comm_with=None):
        # transport: "torch" = torch.distributed point-to-point (RCCL under the nccl backend, gloo on CPU);
        #            "native" = the C-ABI exchange of include/dbcsr_amd_comm.h (RCCL group on a dedicated HIP stream);
        #            "auto"   = native when it can be set up (GPU tensors, more than one rank), else torch
        self.comm = None
        self.transport = "torch"
        if share_comm_with is not None:   # a second plan of the same job (another grid / schedule): the first one's communicator
            self.comm, self.transport = share_comm_with.comm, share_comm_with.transport
        elif transport in ("native", "auto") and dist.is_initialized() and torch.cuda.is_available() and \
                (dist.get_world_size() > 1 or transport == "native"):   # (a one-rank job takes the native path only when told to: tests)
            import sys
            err = None
            try:
                from .comm import NativeComm
                self.comm = NativeComm()
                err = self._native_selftest()
            except Exception as e:  # noqa: BLE001
                err = repr(e)
            # every rank must end up with the same transport: agree over torch's own communicator
            flag_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                self.transport = "native"
            else:
                if self.comm is not None and err is None:
                    self.comm.close()
                self.comm = None   # (a communicator whose self-test did not complete is abandoned, not destroyed: that could block)
                if transport == "native":
                    raise RuntimeError("dbcsr_amd.cannon: native RCCL transport unavailable on some rank (%s)" % (err,))
                sys.stderr.write("dbcsr_amd.cannon: native RCCL transport unavailable (%s), using torch.distributed\n" % (err or "another rank",))
        # ranks of the RCCL communicator the panels travel on (0: no RCCL -- one rank, or the gloo / host-staged debug path)
        self.rccl_ranks = 0
        if dist.is_initialized() and dist.get_world_size() > 1:
            if self.comm is not None:
                self.rccl_ranks = self.comm.world          # as ncclCommInitRank was told (dbcsr_amd_comm_create)
            elif dist.get_backend() == "nccl":
                self.rccl_ranks = dist.get_world_size()
        self.mode = mode
        self.local_first = local_first
        self._engines = {}
        self.last_engine = engine
        self._host = None
        self._owned = None
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        # mode "colpipe": world x 1 grid -- every rank keeps its block rows of A and of C and needs ALL of B, which then travels over
        # all links at once, in column chunks that are multiplied as they arrive (see _multiply_colpipe)
        self.grid = grid or (Grid(world, rank, nprows=world, npcols=1) if mode == "colpipe" else Grid(world, rank))
        # mode "colpipe2d": the same column-chunk pipeline on the default 2-D grid (the A images of the process row come with the first chunk)
        self._col_chunks = max(1, int(col_chunks)) if mode in ("colpipe", "colpipe2d", "tilepipe") else 0
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dtype = dtype
        if engine is None:
            from .multiply import default_engine
            engine = default_engine()
        self.eng = engine
        g = self.grid
        c0 = randmat.RANDMAT_SEED_INIT
        if matrices is not None:
            hA, hB, hC = matrices
            sm, sk, sn = (np.asarray(x, np.int32) for x in (hA.row_sizes, hA.col_sizes, hB.col_sizes))
            rows_of = lambda m: np.repeat(np.arange(len(m.row_sizes), dtype=np.int32), np.diff(np.asarray(m.row_p)))
            self.pat = {w: (rows_of(m), np.asarray(m.col_i, np.int32)) for w, m in (("A", hA), ("B", hB), ("C", hC))}
            self._host = {"A": hA, "B": hB, "C": hC}
        elif distributed is not None:
            # ``distributed=((A_loc, B_loc, C_loc), (row_sizes, k_sizes, col_sizes))``: every rank holds only SOME blocks of
            # each matrix (DistBlocks, global coordinates, any initial ownership); make_images moves them to the owners of
            # their target images and tells every rank which blocks exist (index exchange), see redistribute().
            (dA, dB, dC), (sm, sk, sn) = distributed
            sm, sk, sn = (np.asarray(x, np.int32) for x in (sm, sk, sn))
            P0 = Partition(sm, sk, sn, g)
            # owner of a block's target image (Grid.a_owner / b_owner / rank_of, on whole index arrays)
            dests = {"A": lambda rr, cc: g.a_owner(P0.row_dist[rr].astype(np.int64), P0.k_dist[cc].astype(np.int64)),
                     "B": lambda rr, cc: g.b_owner(P0.k_dist[rr].astype(np.int64), P0.col_dist[cc].astype(np.int64)),
                     "C": lambda rr, cc: g.rank_of(P0.row_dist[rr].astype(np.int64), P0.col_dist[cc].astype(np.int64))}
            self._owned, self.pat = {}, {}
            for w, loc, (rsz, csz) in (("C", dC, (sm, sn)), ("A", dA, (sm, sk)), ("B", dB, (sk, sn))):
                src = loc.data if isinstance(loc.data, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(loc.data))
                # the blocks go to HBM once, before the redistribution: packing, exchange and sorting happen there
                owned, everyone = redistribute(DistBlocks(loc.rows, loc.cols, src.to(self.device, dtype)), dests[w], rsz, csz, dtype, comm=self.comm)
                self._owned[w] = owned
                rows_all = np.concatenate([e[0] for e in everyone]).astype(np.int32)
                cols_all = np.concatenate([e[1] for e in everyone]).astype(np.int32)
                # this rank only ever addresses its process row of A, its process column of B and its own tile of C
                rr, cc = g.myprow, g.mypcol
                if w == "A":
                    keep = P0.row_dist[rows_all] == rr
                elif w == "B":
                    keep = P0.col_dist[cols_all] == cc
                else:
                    keep = (P0.row_dist[rows_all] == rr) & (P0.col_dist[cols_all] == cc)
                order = np.lexsort((cols_all[keep], rows_all[keep]))
                self.pat[w] = (rows_all[keep][order], cols_all[keep][order])
        else:
            sm = randmat.make_random_block_sizes(M, mix)
            sn = randmat.make_random_block_sizes(N, mix_n or mix)
            sk = randmat.make_random_block_sizes(K, mix_k or mix)
            # global patterns, identical on every rank (C, A, B in the reference driver's order)
            self.pat = {"C": randmat.random_pattern(len(sm), len(sn), sparsities[2], c0 + 1),
                        "A": randmat.random_pattern(len(sm), len(sk), sparsities[0], c0 + 2),
                        "B": randmat.random_pattern(len(sk), len(sn), sparsities[1], c0 + 3)}
        self.part = P = Partition(sm, sk, sn, g)
        self.counters = {"C": c0 + 1, "A": c0 + 2, "B": c0 + 3}
        self.nbr_g, self.nbk_g, self.nbc_g = len(sm), len(sk), len(sn)
        r, c = g.myprow, g.mypcol
        self._cbounds = self._rbounds = None
        if self._col_chunks:
            ncl = len(P.cols_of[c])
            nch = max(1, min(self._col_chunks, ncl))
            if mode == "tilepipe":   # as many row chunks of the C tile as column chunks, the same number on every rank (a batch is matched by its peers)
                nch = max(1, min([self._col_chunks] + [len(x) for x in P.cols_of] + [len(x) for x in P.rows_of]))
                self._rbounds = np.round(np.linspace(0, len(P.rows_of[r]), nch + 1)).astype(np.int64)
            self._cbounds = np.round(np.linspace(0, ncl, nch + 1)).astype(np.int64)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(self.device)
        self._rs = t(sm[P.rows_of[r]], torch.int32)
        self._cs = t(sn[P.cols_of[c]], torch.int32)
        self._ks = [t(sk[P.ks_of[v]], torch.int32) for v in range(g.nvirt)]
        self._ks_all = t(sk, torch.int32)
        self._row_gid = t(P.rows_of[r], torch.int32)
        self._col_gid = t(P.cols_of[c], torch.int32)
        self._k_gid = [t(P.ks_of[v], torch.int32) for v in range(g.nvirt)]
        self._k_gid_all = t(np.arange(len(sk), dtype=np.int32), torch.int32)
        # my C tile
        self.C_in = self._make("C", P.row_dist, r, P.col_dist, c, P.row_local, P.col_local, len(P.rows_of[r]), self._rs, self._cs,
                               sm, sn, self._row_gid, self._col_gid, self.nbr_g, fill=True)
        # index of every image this rank will ever multiply with; data only for the owned ones
        self.A_img, self.B_img = {}, {}
        for v in range(g.nvirt):
            own_a = g.a_owner(r, v) == g.rank
            self.A_img[v] = self._make("A", P.row_dist, r, P.k_dist, v, P.row_local, P.k_local, len(P.rows_of[r]), self._rs,
                                       self._ks[v], sm, sk, self._row_gid, self._k_gid[v], self.nbr_g, fill=own_a)
            own_b = g.b_owner(v, c) == g.rank
            self.B_img[v] = self._make("B", P.k_dist, v, P.col_dist, c, P.k_local, P.col_local, len(P.ks_of[v]), self._ks[v],
                                       self._cs, sk, sn, self._k_gid[v], self._col_gid, self.nbk_g, fill=own_b, chunk_bounds=self._cbounds)
        # full-panel patterns (no data) for the one-off symbolic product that fixes C's structure
        all_k_local = np.arange(len(sk), dtype=np.int32)
        self.A_panel = self._make("A", P.row_dist, r, np.zeros(len(sk), np.int32), 0, P.row_local, all_k_local, len(P.rows_of[r]),
                                  self._rs, self._ks_all, sm, sk, None, None, 0, fill=None)
        self.B_panel = self._make("B", np.zeros(len(sk), np.int32), 0, P.col_dist, c, all_k_local, P.col_local, len(sk), self._ks_all,
                                  self._cs, sk, sn, None, None, 0, fill=None)
        # gather mode: one buffer per panel = the images back to back; the panel index points into it
        self._a_base, self._b_base, oa, ob = {}, {}, 0, 0
        for v in range(g.nvirt):
            self._a_base[v], self._b_base[v] = oa, ob
            oa += self.A_img[v].data_numel
            ob += self.B_img[v].data_numel
        self._a_all = torch.empty(oa, dtype=dtype, device=self.device)
        self._b_all = torch.empty(ob, dtype=dtype, device=self.device)
        self.A_panel.blk_p = self._panel_blk_p("A", P.row_dist, r, None, P.k_dist, self._a_base, self.A_img, P.row_local, P.k_local, rows_are_k=False)
        self.B_panel.blk_p = self._panel_blk_p("B", None, None, (P.col_dist, c), P.k_dist, self._b_base, self.B_img, P.k_local, P.col_local, rows_are_k=True)
        self.A_panel.data, self.B_panel.data = self._a_all, self._b_all
        for v in range(g.nvirt):  # owned images live inside the panel buffers (no copy at multiply time)
            for img, base, buf in ((self.A_img[v], self._a_base[v], self._a_all), (self.B_img[v], self._b_base[v], self._b_all)):
                if img.data.numel():
                    buf[base:base + img.data_numel].copy_(img.data)
                    img.data = buf[base:base + img.data_numel]
        # local-first split of the gather schedule: images this rank owns on BOTH sides can be multiplied while
        # the other images are still travelling
        self._S1 = [v for v in range(g.nvirt) if g.a_owner(r, v) == g.rank and g.b_owner(v, c) == g.rank]
        self._S2 = [v for v in range(g.nvirt) if v not in self._S1]
        self._sub = {}
        for name, S in (("S1", self._S1), ("S2", self._S2)):
            self._sub[name] = (self._sub_panel("A", S), self._sub_panel("B", S)) if S and g.world > 1 else None
        # colpipe: B's column panel and C_in cut into column chunks (same buffers, chunk-local column numbering)
        self._Bc = self._Cc = None
        self._merged = None
        self.colpipe_copies = 0
        self.colpipe_two_streams = os.environ.get("DBCSR_AMD_COLPIPE_STREAMS", "2") != "1"
        self._side_stream = self._arrival_stream = None
        self._origins = None   # (first block row, first block column) of every part of C a step produces
        if self._rbounds is not None:
            # tilepipe: step s = the row strip (row chunk s) x (column chunks 0 .. s), then the column strip (row chunks 0 .. s - 1) x (column chunk s)
            rb, cb, nk = self._rbounds, self._cbounds, len(sk)
            self._tiles = []
            for q in range(len(rb) - 1):
                self._tiles.append((q, int(rb[q]), int(rb[q + 1]), 0, int(cb[q + 1])))
                if q:
                    self._tiles.append((q, 0, int(rb[q]), int(cb[q]), int(cb[q + 1])))
            self._tiles = [t for t in self._tiles if t[2] > t[1] and t[4] > t[3]]
            self._At = [self._sub_rc(self.A_panel, rlo, rhi, 0, nk) for _, rlo, rhi, _, _ in self._tiles]
            self._Bc = [self._col_sub(self.B_panel, clo, chi) for _, _, _, clo, chi in self._tiles]
            self._Cc = [self._sub_rc(self.C_in, rlo, rhi, clo, chi) for _, rlo, rhi, clo, chi in self._tiles]
            self._origins = [(rlo, clo) for _, rlo, _, clo, _ in self._tiles]
            for v in range(g.nvirt):   # an A image is stored by (row, column): a row chunk of it is one piece
                img = self.A_img[v]
                rp = img.row_p.detach().cpu().numpy().astype(np.int64)
                bp = np.append(img.blk_p.detach().cpu().numpy().astype(np.int64), img.data_numel)
                img.row_chunk_off = bp[rp[rb]]
        elif self._cbounds is not None:
            nch = len(self._cbounds) - 1
            self._Bc = [self._col_sub(self.B_panel, int(self._cbounds[q]), int(self._cbounds[q + 1])) for q in range(nch)]
            self._Cc = [self._col_sub(self.C_in, int(self._cbounds[q]), int(self._cbounds[q + 1])) for q in range(nch)]
            self._origins = [(0, int(self._cbounds[q])) for q in range(nch)]
        # double-buffered receive space for A and B panels
        amax = max([m.data_numel for v, m in self.A_img.items() if g.a_owner(r, v) != g.rank] + [0])
        bmax = max([m.data_numel for v, m in self.B_img.items() if g.b_owner(v, c) != g.rank] + [0])
        self._abuf = [torch.empty(amax, dtype=dtype, device=self.device) for _ in range(2)]
        self._bbuf = [torch.empty(bmax, dtype=dtype, device=self.device) for _ in range(2)]

    def _engine(self, key):
        """The local multiply engine for one of the multiplies of a step.  An engine keeps the plan (symbolic product, product lists,
        launch order) of its LAST multiply and reuses it when the next one has the same index arrays (include/dbcsr_amd_mm.h: plan
        reuse) -- which is the case for every repetition of a distributed multiply, per tick / per part of the gather schedule.  So
        each of them gets an engine of its own; `None` is the step's first multiply (the caller's engine).
        Device memory: an engine's workspace is the plan of ITS multiply -- 12 bytes per block product, 84 per C block (descriptor,
        work record, launch order) and the bitmaps of its operands' index -- and the ticks / parts / column chunks of a step partition
        the step's products, so all engines of a rank together hold what ONE engine would hold for the rank's whole share
        (config 2 on one rank: 207 M products -> 2.5 GB; on eight: 0.3 GB per rank).  The number of engines is nvirt + column chunks,
        fixed when the plan object is built."""
        if key is None or not hasattr(self.eng, "plan_stats"):
            return self.eng
        e = self._engines.get(key)
        if e is None:
            e = self._engines[key] = type(self.eng)(lab=self.eng.lab) if hasattr(self.eng, 'lab') else type(self.eng)()
            if hasattr(e, "trust_plan"):
                e.trust_plan(True)   # the plan's operands are this object's own and their index is never written again
        return e

    def _native_selftest(self, timeout_s=60.0):
        """One tiny ring exchange through the native transport before it is trusted with the panels; returns None or what went
        wrong.  The wait is bounded: a transport that does not complete here falls back to torch.distributed instead of hanging
        the first multiply."""
        import time
        c = self.comm
        nxt, prv = (c.rank + 1) % c.world, (c.rank - 1) % c.world
        out = torch.full((16,), float(c.rank), dtype=torch.float64, device="cuda")
        inp = torch.full((16,), -1.0, dtype=torch.float64, device="cuda")
        ev = c.exchange([(out, nxt)], [(inp, prv)])
        t0 = time.perf_counter()
        while not ev.query():
            if time.perf_counter() - t0 > timeout_s:
                return "self-test exchange did not complete within %.0f s" % timeout_s
            time.sleep(0.001)
        if not bool(torch.all(inp == float(prv))):
            return "self-test exchange delivered wrong data"
        return None

    def _make(self, which, rdist, rsel, cdist, csel, rloc, cloc, nrows_local, rs_t, cs_t, rsizes, csizes, rgid, cgid, nrow_global, fill,
              chunk_bounds=None):
        rows, cols = self.pat[which]
        keep = (rdist[rows] == rsel) & (cdist[cols] == csel)
        row_p, col_i, blk_p, nze = _sub_index(rows, cols, keep, rloc, cloc, nrows_local, rsizes, csizes)
        chunk_off = order = None
        if chunk_bounds is not None:
            # data laid out column chunk by column chunk (inside a chunk: by (row, col) as before), so that the blocks of one chunk
            # are one contiguous piece of the image -- what travels per step of the colpipe schedule
            sizes = np.diff(np.append(blk_p, nze))
            chunk = np.searchsorted(chunk_bounds, col_i, side="right") - 1
            order = np.argsort(chunk, kind="stable")
            starts = np.zeros(len(order), np.int64)
            if len(order):
                starts[1:] = np.cumsum(sizes[order])[:-1]
            blk_p = np.empty_like(blk_p)
            blk_p[order] = starts
            per_chunk = np.bincount(chunk, weights=sizes, minlength=len(chunk_bounds) - 1).astype(np.int64) if len(order) else \
                np.zeros(len(chunk_bounds) - 1, np.int64)
            chunk_off = np.concatenate([[0], np.cumsum(per_chunk)])
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(self.device)
        data = torch.empty(nze if fill else 0, dtype=self.dtype, device=self.device)
        M = DbcsrMatrix(rs_t, cs_t, t(row_p, torch.int32), t(col_i, torch.int32), t(blk_p, torch.int64), data, which)
        M.data_numel = nze
        M.chunk_off = chunk_off
        if fill and nze:
            if self._owned is not None:  # the blocks this rank received in make_images (sorted by global (row, col))
                orow, ocol, ooff, odat = self._owned[which]
                ncol_t = len(csizes)
                okey = orow.astype(np.int64) * ncol_t + ocol
                kkey = rows[keep].astype(np.int64) * ncol_t + cols[keep]
                pos = np.searchsorted(okey, kkey)
                assert len(okey) and np.array_equal(okey[pos], kkey), "image block missing after redistribution"
                if order is not None:
                    pos = pos[order]   # (the blocks in the order they are stored)
                M.data.copy_(gather_blocks(odat, ooff[pos], ooff[pos + 1] - ooff[pos]).to(self.device))
            elif self._host is not None:  # cut the kept blocks out of the replicated global matrix
                h = self._host[which]
                hb, hd = np.asarray(h.blk_p, np.int64), np.asarray(h.data)
                sizes = rsizes[rows[keep]].astype(np.int64) * csizes[cols[keep]].astype(np.int64)
                src = hb[np.nonzero(keep)[0]]
                if order is not None:
                    src, sizes = src[order], sizes[order]
                M.data.copy_(torch.as_tensor(np.concatenate([hd[o:o + n] for o, n in zip(src, sizes)])).to(self.device))
            else:
                self.eng.fill_random_dist(M, self.counters[which], rgid, cgid, nrow_global)
        return M

    def _panel_blk_p(self, which, rdist, rsel, csel, k_dist, bases, imgs, rloc, cloc, rows_are_k):
        """Offsets of the panel's blocks inside the concatenated image buffer (image base + offset in image)."""
        rows, cols = self.pat[which]
        if rows_are_k:   # B panel: rows are k blocks (image = k_dist[row]), columns restricted to my process column
            cdist, cc = csel
            keep = cdist[cols] == cc
            img_of = k_dist[rows[keep]]
        else:            # A panel: rows restricted to my process row, columns are k blocks (image = k_dist[col])
            keep = rdist[rows] == rsel
            img_of = k_dist[cols[keep]]
        out = np.zeros(int(keep.sum()), np.int64)
        kept = np.nonzero(keep)[0]
        for v, img in imgs.items():
            sel = img_of == v
            blk = img.blk_p.detach().cpu().numpy()
            assert int(sel.sum()) == len(blk)
            out[sel] = bases[v] + blk  # both enumerate the image's blocks in global (row, col) order
        return torch.as_tensor(out, dtype=torch.int64).to(self.device)

    def _sub_panel(self, which, S):
        """Panel restricted to the k-images in S: same buffers and global k numbering, fewer blocks."""
        P, g = self.part, self.grid
        full = self.A_panel if which == "A" else self.B_panel
        rs, cs, row_p, col_i, blk_p, _ = [x for x in DbcsrMatrix(full.row_blk_size, full.col_blk_size, full.row_p, full.col_i, full.blk_p,
                                                                full.row_p[:0]).to_host()]
        rows = np.repeat(np.arange(len(rs), dtype=np.int64), np.diff(row_p))
        kk = col_i if which == "A" else rows            # the k index of each block (global numbering)
        keep = np.isin(P.k_dist[kk], np.asarray(S, np.int32))
        nrow_p = np.zeros(len(rs) + 1, np.int64)
        np.add.at(nrow_p, rows[keep] + 1, 1)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(self.device)
        M = DbcsrMatrix(full.row_blk_size, full.col_blk_size, t(np.cumsum(nrow_p), torch.int32), t(col_i[keep], torch.int32),
                        t(blk_p[keep], torch.int64), full.data, which)
        return M

    def _col_sub(self, full, lo, hi):
        """`full` restricted to its block columns lo .. hi - 1, numbered from 0: same data buffer, fewer blocks."""
        rs, cs, row_p, col_i, blk_p, _ = DbcsrMatrix(full.row_blk_size, full.col_blk_size, full.row_p, full.col_i, full.blk_p,
                                                     full.row_p[:0]).to_host()
        rows = np.repeat(np.arange(len(rs), dtype=np.int64), np.diff(row_p))
        keep = (col_i >= lo) & (col_i < hi)
        nrow_p = np.zeros(len(rs) + 1, np.int64)
        np.add.at(nrow_p, rows[keep] + 1, 1)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(self.device)
        return DbcsrMatrix(full.row_blk_size, full.col_blk_size[lo:hi], t(np.cumsum(nrow_p), torch.int32), t(col_i[keep] - lo, torch.int32),
                           t(blk_p[keep], torch.int64), full.data, full.name)

    def _sub_rc(self, full, rlo, rhi, clo, chi):
        """`full` restricted to its block rows rlo .. rhi - 1 and block columns clo .. chi - 1, both numbered from 0: same data buffer."""
        rs, cs, row_p, col_i, blk_p, _ = DbcsrMatrix(full.row_blk_size, full.col_blk_size, full.row_p, full.col_i, full.blk_p,
                                                     full.row_p[:0]).to_host()
        rows = np.repeat(np.arange(len(rs), dtype=np.int64), np.diff(row_p))
        keep = (col_i >= clo) & (col_i < chi) & (rows >= rlo) & (rows < rhi)
        nrow_p = np.zeros(rhi - rlo + 1, np.int64)
        np.add.at(nrow_p, rows[keep] - rlo + 1, 1)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(self.device)
        return DbcsrMatrix(full.row_blk_size[rlo:rhi], full.col_blk_size[clo:chi], t(np.cumsum(nrow_p), torch.int32), t(col_i[keep] - clo, torch.int32),
                           t(blk_p[keep], torch.int64), full.data, full.name)

    def _exchange(self, sends, recvs):
        """Posts one batch of (tensor, peer) sends and receives on the transport in use; returns (work handles, staged host copies)."""
        g = self.grid
        if not sends and not recvs:
            return [], []
        if self.comm is not None:  # native transport: ONE RCCL group on the communication stream
            return [_EventWork(self.comm.exchange(sends, recvs))], []
        ops, staged = [], []
        host = g.world > 1 and self.device.type == "cuda" and dist.get_backend() != "nccl"  # debug transport
        if host:
            torch.cuda.synchronize()
        for t, peer in sends:
            if host:
                t = t.cpu()
                staged.append(t)
            ops.append(dist.P2POp(dist.isend, t, peer))
        for t, peer in recvs:
            if host:
                h = torch.empty(t.numel(), dtype=t.dtype)
                staged.append((h, t))
                ops.append(dist.P2POp(dist.irecv, h, peer))
            else:
                ops.append(dist.P2POp(dist.irecv, t, peer))
        return dist.batch_isend_irecv(ops), staged

    @staticmethod
    def _arrived(works, staged):
        for w in works:
            w.wait()
        for item in staged:
            if isinstance(item, tuple):
                item[1].copy_(item[0])

    def _launched(self, eng, flop):
        """Book-keeping of one local block-product launch of the step: what bench.py's roofline refers to.  With
        collect_kernel_times set (bench.py's untimed roofline steps) the launch's HIP-event time is read back -- a synchronisation --
        and kept in step_launches as (kernel ms, flop), so that a step's kernel time is the SUM over its chunks / ticks / parts."""
        self.last_tick_flop = flop
        if getattr(self, "collect_kernel_times", False):
            passes = getattr(eng, "pass_launches", None) if getattr(eng, "last_kchunks", 1) > 1 else None
            if passes:   # k passes: one launch per pass (multiply.py)
                self.step_launches.extend(passes)
            else:
                self.step_launches.append((float(eng.last_timing()[1]), int(flop)))

    def _multiply_colpipe(self, alpha, beta):
        """world x 1 grid.  A's block rows are local; B's column panel is ALL of B, one k-image per rank.  It travels in column chunks
        -- chunk q of every image, from its owner to everybody, one batch per chunk, all batches posted at once and carried out in
        order -- and chunk q of C (disjoint block columns: every C block is written once) is multiplied as soon as chunk q of B is
        complete, while chunk q + 1 is on the links.  Exposed transfer: one chunk of one image per link instead of a whole image;
        the k-sliced tick schedule overlaps as much but reads and writes C once per tick."""
        g, r, c = self.grid, self.grid.myprow, self.grid.mypcol
        nch = len(self._cbounds) - 1
        posted = []
        for q in range(nch):
            sends, recvs = [], []
            for v in range(g.nvirt):
                if q == 0:  # (grids with more than one process column: the A images this rank misses come with the first chunk)
                    na, a_own = self.A_img[v].data_numel, g.a_owner(r, v)
                    if na and a_own == g.rank:
                        sends += [(self.A_img[v].data, g.rank_of(r, pc)) for pc in range(g.npcols) if pc != c]
                    elif na:
                        recvs.append((self._a_all[self._a_base[v]:self._a_base[v] + na], a_own))
                img, b_own = self.B_img[v], g.b_owner(v, c)
                lo, hi = int(img.chunk_off[q]), int(img.chunk_off[q + 1])
                if hi > lo:
                    if b_own == g.rank:
                        sends += [(img.data[lo:hi], g.rank_of(pr, c)) for pr in range(g.nprows) if pr != r]
                    else:
                        recvs.append((self._b_all[self._b_base[v] + lo:self._b_base[v] + hi], b_own))
            posted.append(self._exchange(sends, recvs))
        engines = [self._engine(("col", q)) for q in range(nch)]   # (none of them the caller's: these reuse their plans by address)
        # C's structure per chunk needs the (replicated) index only: the symbolic phases run while the first chunk travels
        sym = [engines[q].symbolic(self.A_panel, self._Bc[q], self._Cc[q], retain_sparsity=False) for q in range(nch)]
        mg = self._merged
        out_all = torch.empty(mg["nze"], dtype=self.dtype, device=self.device) if mg is not None else None
        parts, flop, nprod = [], 0, 0
        # Odd chunks are multiplied on a second stream: a chunk's kernel is short (0.7 ms at 8 ranks), and the tail of one then runs
        # under the head of the next (they touch disjoint C blocks and have an engine each; each waits for its own batch only)
        main = side = None
        if self.device.type == "cuda" and nch > 1 and self.colpipe_two_streams:
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=self.device)
            side = self._side_stream
            side.wait_stream(main)   # the symbolic phases' results, and whatever used the output buffer's memory before
            # (no record_stream on the output buffer: the first stream waits for the second below, before anybody can free it --
            #  and a recorded buffer of several GB is not reusable by the next step's allocation, which then goes to hipMalloc)
        a_landed = None
        for q in range(nch):
            row_p, cnt = sym[q]
            eng = self.last_engine = engines[q]
            kw = {}
            if out_all is not None and getattr(eng, "accepts_out_data", False):
                kw["out_data"] = out_all[mg["off"][q]:mg["off"][q] + cnt.c_nze]
            on_side = side is not None and q % 2 == 1
            with torch.cuda.stream(side if on_side else main) if main is not None else contextlib.nullcontext():
                if on_side and a_landed is not None:
                    side.wait_event(a_landed)   # grids with several process columns: the A images came with batch 0, which the first stream took in
                self._arrived(*posted[q])
                if q == 0 and side is not None and g.npcols > 1:
                    a_landed = torch.cuda.Event()
                    a_landed.record(main)
                parts.append(eng.numeric_after_symbolic(alpha, self.A_panel, self._Bc[q], beta, self._Cc[q], row_p, cnt, self.dtype, **kw))
            flop += cnt.flop
            nprod += cnt.nproducts
            self._launched(eng, cnt.flop)
        if side is not None:
            main.wait_stream(side)
            for Cq in parts[1::2]:   # made on the second stream, used by the caller on the first
                for t in (Cq.col_i, Cq.blk_p) + ((Cq.data,) if out_all is None else ()):   # (data: own allocations in a plan's first multiply only)
                    if t.numel():
                        t.record_stream(main)
        counts = sym[0][1]
        counts.flop, counts.nproducts = flop, nprod
        counts.c_nblks, counts.c_nze = sum(int(x[1].c_nblks) for x in sym), sum(int(x[1].c_nze) for x in sym)
        return self._merge_chunks(parts, out_all), counts

    def _multiply_tilepipe(self, alpha, beta):
        """2-D grid.  Batch s carries row chunk s of every A image this rank misses (from the owners in its process row) and column chunk
        s of every B image it misses (from the owners in its process column); all batches are posted at once and carried out in order.
        Step s multiplies what batch s made computable: (row chunk s) x (column chunks 0 .. s) on the first stream and
        (row chunks 0 .. s - 1) x (column chunk s) on the second.  Every C block is written once, into its slice of one buffer."""
        g, r, c = self.grid, self.grid.myprow, self.grid.mypcol
        nch = len(self._cbounds) - 1
        posted = []
        for q in range(nch):
            sends, recvs = [], []
            for v in range(g.nvirt):
                img, a_own = self.A_img[v], g.a_owner(r, v)
                lo, hi = int(img.row_chunk_off[q]), int(img.row_chunk_off[q + 1])
                if hi > lo:
                    if a_own == g.rank:
                        sends += [(img.data[lo:hi], g.rank_of(r, pc)) for pc in range(g.npcols) if pc != c]
                    else:
                        recvs.append((self._a_all[self._a_base[v] + lo:self._a_base[v] + hi], a_own))
                img, b_own = self.B_img[v], g.b_owner(v, c)
                lo, hi = int(img.chunk_off[q]), int(img.chunk_off[q + 1])
                if hi > lo:
                    if b_own == g.rank:
                        sends += [(img.data[lo:hi], g.rank_of(pr, c)) for pr in range(g.nprows) if pr != r]
                    else:
                        recvs.append((self._b_all[self._b_base[v] + lo:self._b_base[v] + hi], b_own))
            posted.append(self._exchange(sends, recvs))
        nt = len(self._tiles)
        engines = [self._engine(("tile", i)) for i in range(nt)]
        # the structure of every strip of C needs the (replicated) index only: the symbolic phases run while the first batch travels
        sym = [engines[i].symbolic(self._At[i], self._Bc[i], self._Cc[i], retain_sparsity=False) for i in range(nt)]
        mg = self._merged
        out_all = torch.empty(mg["nze"], dtype=self.dtype, device=self.device) if mg is not None else None
        main = side = arr = None
        if self.device.type == "cuda" and nt > 1 and self.colpipe_two_streams:
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=self.device)
            if self._arrival_stream is None:
                self._arrival_stream = torch.cuda.Stream(device=self.device)
            side, arr = self._side_stream, self._arrival_stream
            side.wait_stream(main)   # the symbolic phases' results, and whatever used the output buffer's memory before
            arr.wait_stream(main)
        landed = []   # batch q has arrived: taken in on a stream of its own, so that neither multiply stream waits for the other's kernels
        for q in range(nch):
            if arr is not None:
                with torch.cuda.stream(arr):
                    self._arrived(*posted[q])
                    ev = torch.cuda.Event()
                    ev.record(arr)
                landed.append(ev)
            else:
                landed.append(None)
        taken = -1
        parts, flop, nprod = [], 0, 0
        for i, (q, rlo, rhi, clo, chi) in enumerate(self._tiles):
            row_p, cnt = sym[i]
            eng = self.last_engine = engines[i]
            kw = {}
            if out_all is not None and getattr(eng, "accepts_out_data", False):
                kw["out_data"] = out_all[mg["off"][i]:mg["off"][i] + cnt.c_nze]
            on_side = side is not None and clo > 0   # the column strips
            with torch.cuda.stream(side if on_side else main) if main is not None else contextlib.nullcontext():
                if arr is not None:
                    (side if on_side else main).wait_event(landed[q])
                else:
                    while taken < q:
                        taken += 1
                        self._arrived(*posted[taken])
                parts.append(eng.numeric_after_symbolic(alpha, self._At[i], self._Bc[i], beta, self._Cc[i], row_p, cnt, self.dtype, **kw))
            flop += cnt.flop
            nprod += cnt.nproducts
            self._launched(eng, cnt.flop)
        if arr is None:
            while taken < nch - 1:   # (a batch no strip waited for: empty strips)
                taken += 1
                self._arrived(*posted[taken])
        if side is not None:
            main.wait_stream(side)
            main.wait_stream(arr)
            for i, Cq in enumerate(parts):
                if self._tiles[i][3] > 0:   # made on the second stream, used by the caller on the first
                    for t in (Cq.col_i, Cq.blk_p) + ((Cq.data,) if out_all is None else ()):
                        if t.numel():
                            t.record_stream(main)
        counts = sym[0][1]
        counts.flop, counts.nproducts = flop, nprod
        counts.c_nblks, counts.c_nze = sum(int(x[1].c_nblks) for x in sym), sum(int(x[1].c_nze) for x in sym)
        return self._merge_chunks(parts, out_all), counts

    def _merge_chunks(self, parts, out_all):
        """The column chunks (tilepipe: the strips) of C as ONE matrix of the local tile.  The patterns of a plan never change, so the merged index is made
        once; from the second multiply on the chunks were written straight into their slices of one buffer (no copy)."""
        if self._merged is None:
            rows_l, cols_l, off_l, base = [], [], [], 0
            offs = []
            for q, Cq in enumerate(parts):
                rs, cs, row_p, col_i, blk_p, _ = DbcsrMatrix(Cq.row_blk_size, Cq.col_blk_size, Cq.row_p, Cq.col_i, Cq.blk_p, Cq.row_p[:0]).to_host()
                rows = np.repeat(np.arange(len(rs), dtype=np.int64), np.diff(row_p))
                rows_l.append(rows + self._origins[q][0])
                cols_l.append(col_i.astype(np.int64) + self._origins[q][1])
                off_l.append(blk_p.astype(np.int64) + base)
                offs.append(base)
                base += int(Cq.data.numel())
            rows, cols, off = np.concatenate(rows_l), np.concatenate(cols_l), np.concatenate(off_l)
            order = np.lexsort((cols, rows))
            row_p = np.zeros(len(self._rs) + 1, np.int64)
            np.add.at(row_p, rows + 1, 1)
            t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(self.device)
            self._merged = {"row_p": t(np.cumsum(row_p), torch.int32), "col_i": t(cols[order], torch.int32), "blk_p": t(off[order], torch.int64),
                            "off": offs, "nze": base, "sizes": [int(Cq.data.numel()) for Cq in parts]}
        mg = self._merged
        assert [int(Cq.data.numel()) for Cq in parts] == mg["sizes"], "the pattern of a chunk of C changed between two multiplies of one plan"
        if out_all is None or any(Cq.data.data_ptr() != out_all[mg["off"][q]:].data_ptr() for q, Cq in enumerate(parts) if Cq.data.numel()):
            out_all = torch.cat([Cq.data for Cq in parts]) if parts else torch.empty(0, dtype=self.dtype, device=self.device)
            self.colpipe_copies += 1   # (the first multiply of a plan, or an engine that allocates its own output)
        return DbcsrMatrix(self._rs, self._cs, mg["row_p"], mg["col_i"], mg["blk_p"], out_all, "C")

    def _post_all(self):
        """gather mode: one batch with every image this rank misses (and every send the others expect)."""
        g, r, c = self.grid, self.grid.myprow, self.grid.mypcol
        ops, staged = [], []
        host = self.comm is None and g.world > 1 and self.device.type == "cuda" and dist.get_backend() != "nccl"  # debug transport
        if host:
            torch.cuda.synchronize()

        def send(t, peer):
            if host:
                t = t.cpu()
                staged.append(t)
            ops.append(dist.P2POp(dist.isend, t, peer))

        def recv(t, peer):
            if host:
                h = torch.empty(t.numel(), dtype=t.dtype)
                staged.append((h, t))
                ops.append(dist.P2POp(dist.irecv, h, peer))
            else:
                ops.append(dist.P2POp(dist.irecv, t, peer))

        nsends, nrecvs = [], []
        if self.comm is not None:  # native transport: collect (tensor, peer), post ONE RCCL group on the communication stream
            send = lambda t, peer: nsends.append((t, peer))
            recv = lambda t, peer: nrecvs.append((t, peer))
        for v in range(g.nvirt):
            na, nb = self.A_img[v].data_numel, self.B_img[v].data_numel
            a_own, b_own = g.a_owner(r, v), g.b_owner(v, c)
            if na:
                if a_own == g.rank:
                    for pc in range(g.npcols):
                        if pc != c:
                            send(self.A_img[v].data, g.rank_of(r, pc))
                else:
                    recv(self._a_all[self._a_base[v]:self._a_base[v] + na], a_own)
            if nb:
                if b_own == g.rank:
                    for pr in range(g.nprows):
                        if pr != r:
                            send(self.B_img[v].data, g.rank_of(pr, c))
                else:
                    recv(self._b_all[self._b_base[v]:self._b_base[v] + nb], b_own)
        if self.comm is not None:
            return ([_EventWork(self.comm.exchange(nsends, nrecvs))] if (nsends or nrecvs) else []), staged
        works = dist.batch_isend_irecv(ops) if ops else []
        return works, staged

    def comm_probe(self, reps=2):
        """The step's panel exchange ALONE (no multiply beside it): every image this rank misses travels as in one step of the gather
        schedule, timed on the host between two device synchronisations (the caller reduces over the ranks).  Returns the bytes this
        rank received, the largest share that came from ONE peer -- over ONE xGMI link -- and the best time: what bench.py's
        `comm` object (GB/s per link, overlap fraction) is made of."""
        import time
        g, r, c = self.grid, self.grid.myprow, self.grid.mypcol
        esz = torch.empty(0, dtype=self.dtype).element_size()
        per_peer = {}
        for v in range(g.nvirt):
            for n, own in ((self.A_img[v].data_numel, g.a_owner(r, v)), (self.B_img[v].data_numel, g.b_owner(v, c))):
                if n and own != g.rank:
                    per_peer[own] = per_peer.get(own, 0) + n * esz
        best = float("inf")
        sync = torch.cuda.synchronize if self.device.type == "cuda" else (lambda: None)
        for _ in range(max(1, reps)):
            sync()
            if dist.is_initialized() and g.world > 1:
                dist.barrier()
            sync()
            t0 = time.perf_counter()
            works, staged = self._post_all()
            self._arrived(works, staged)
            sync()
            best = min(best, time.perf_counter() - t0)
        return {"bytes_in": int(sum(per_peer.values())), "peers_in": len(per_peer), "max_bytes_from_one_peer": int(max(per_peer.values()) if per_peer else 0),
                "ms": best * 1e3}

    def _multiply_gather(self, alpha, beta):
        eng = self.last_engine = self.eng
        works, staged = self._post_all()                       # panels travel over all links ...

        def arrived():
            for w in works:
                w.wait()
            for item in staged:
                if isinstance(item, tuple):
                    item[1].copy_(item[0])

        s1, s2 = self._sub["S1"], self._sub["S2"]
        if s1 is None or s2 is None or not self.local_first:
            # ... during the symbolic phase of the one multiply over the full panels
            auto_k = getattr(eng, "_auto_kchunks", None)
            if auto_k is not None and auto_k(self.A_panel, 0.0) > 1:  # large A block-rows: passes over k (multiply.py)
                arrived()
                Cout, counts = eng.multiply_local(alpha, self.A_panel, self.B_panel, beta, self.C_in)
                self._launched(eng, getattr(eng, "last_launch_flop", counts.flop))
                return Cout, counts
            row_p, counts = eng.symbolic(self.A_panel, self.B_panel, self.C_in, retain_sparsity=False)
            arrived()
            Cout = eng.numeric_after_symbolic(alpha, self.A_panel, self.B_panel, beta, self.C_in, row_p, counts, self.dtype)
            self._launched(eng, counts.flop)
            return Cout, counts
        # local-first: the images owned on both sides are multiplied while the rest is still in flight
        C1, cnt1 = eng.multiply_local(alpha, s1[0], s1[1], beta, self.C_in)
        self._launched(eng, getattr(eng, "last_launch_flop", cnt1.flop))
        eng = self.last_engine = self._engine("gather-2")   # the second part keeps its own plan
        auto_k = getattr(eng, "_auto_kchunks", None)
        if auto_k is not None and auto_k(s2[0], 0.0) > 1:  # large A block-rows: passes over k (multiply.py)
            arrived()
            Cout, cnt2 = eng.multiply_local(alpha, s2[0], s2[1], 1.0, C1)
            self._launched(eng, getattr(eng, "last_launch_flop", cnt2.flop))
        else:
            # the symbolic phase of the second part needs only the (replicated) index: it runs while the panels still travel
            row_p, cnt2 = eng.symbolic(s2[0], s2[1], C1, retain_sparsity=False)
            arrived()
            Cout = eng.numeric_after_symbolic(alpha, s2[0], s2[1], 1.0, C1, row_p, cnt2, self.dtype)
            self._launched(eng, cnt2.flop)
        cnt2.flop += cnt1.flop
        cnt2.nproducts += cnt1.nproducts
        return Cout, cnt2

    # ------------------------------------------------------------------
    def _post(self, tick, parity):
        """Send/receive the panels of `tick`.  Returns (ops work handles, A data, B data)."""
        if self.comm is None and self.grid.world > 1 and self.device.type == "cuda" and dist.get_backend() != "nccl":
            return self._post_host_staged(tick, parity)
        g, r, c = self.grid, self.grid.myprow, self.grid.mypcol
        ops = []
        nsends, nrecvs = [], []
        v = g.v_at(r, c, tick)
        a_src, b_src = g.a_owner(r, v), g.b_owner(v, c)
        a_data = self.A_img[v].data if a_src == g.rank else self._abuf[parity][:self.A_img[v].data_numel]
        b_data = self.B_img[v].data if b_src == g.rank else self._bbuf[parity][:self.B_img[v].data_numel]
        # what the others need from me at this tick
        for pc in range(g.npcols):  # my process row: A images I own
            if pc == c:
                continue
            vv = g.v_at(r, pc, tick)
            if g.a_owner(r, vv) == g.rank and self.A_img[vv].data_numel:
                nsends.append((self.A_img[vv].data, g.rank_of(r, pc)))
        for pr in range(g.nprows):  # my process column: B images I own
            if pr == r:
                continue
            vv = g.v_at(pr, c, tick)
            if g.b_owner(vv, c) == g.rank and self.B_img[vv].data_numel:
                nsends.append((self.B_img[vv].data, g.rank_of(pr, c)))
        if a_src != g.rank and self.A_img[v].data_numel:
            nrecvs.append((a_data, a_src))
        if b_src != g.rank and self.B_img[v].data_numel:
            nrecvs.append((b_data, b_src))
        if self.comm is not None:  # one RCCL group on the communication stream; it first waits for the compute stream, so the
            # receive buffer of this parity is no longer read by the multiply of two ticks ago
            works = [_EventWork(self.comm.exchange(nsends, nrecvs))] if (nsends or nrecvs) else []
            return works, v, a_data, b_data
        ops = [dist.P2POp(dist.isend, t, p) for t, p in nsends] + [dist.P2POp(dist.irecv, t, p) for t, p in nrecvs]
        works = dist.batch_isend_irecv(ops) if ops else []
        return works, v, a_data, b_data

    def _post_host_staged(self, tick, parity):
        """Debug transport (non-RCCL backends with device tensors, e.g. several ranks sharing one GPU under
        gloo): same schedule, payloads staged through host memory, blocking.  Not a performance path."""
        g, r, c = self.grid, self.grid.myprow, self.grid.mypcol
        v = g.v_at(r, c, tick)
        a_src, b_src = g.a_owner(r, v), g.b_owner(v, c)
        a_data = self.A_img[v].data if a_src == g.rank else self._abuf[parity][:self.A_img[v].data_numel]
        b_data = self.B_img[v].data if b_src == g.rank else self._bbuf[parity][:self.B_img[v].data_numel]
        torch.cuda.synchronize()
        ops, keep, recvs = [], [], []
        for pc in range(g.npcols):
            vv = g.v_at(r, pc, tick)
            if pc != c and g.a_owner(r, vv) == g.rank and self.A_img[vv].data_numel:
                keep.append(self.A_img[vv].data.cpu())
                ops.append(dist.P2POp(dist.isend, keep[-1], g.rank_of(r, pc)))
        for pr in range(g.nprows):
            vv = g.v_at(pr, c, tick)
            if pr != r and g.b_owner(vv, c) == g.rank and self.B_img[vv].data_numel:
                keep.append(self.B_img[vv].data.cpu())
                ops.append(dist.P2POp(dist.isend, keep[-1], g.rank_of(pr, c)))
        if a_src != g.rank and self.A_img[v].data_numel:
            recvs.append((torch.empty(a_data.numel(), dtype=self.dtype), a_data))
            ops.append(dist.P2POp(dist.irecv, recvs[-1][0], a_src))
        if b_src != g.rank and self.B_img[v].data_numel:
            recvs.append((torch.empty(b_data.numel(), dtype=self.dtype), b_data))
            ops.append(dist.P2POp(dist.irecv, recvs[-1][0], b_src))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        for host, dev in recvs:
            dev.copy_(host)
        return [], v, a_data, b_data

    def multiply(self, alpha=1.0, beta=1.0, retain_sparsity=False, filter_eps=None):
        """One distributed multiply; returns (local C_out, counts with this rank's flop).
        retain_sparsity / filter_eps as in dbcsr_multiply.  The on-the-fly filter compares every product with
        (filter_eps / number of blocks of the WHOLE block row of A)^2 (the reference sums the row counts over the process row first,
        dbcsr_mm_cannon.F:1040-1113): such a multiply runs over the full row / column panels in one piece -- every rank then takes
        exactly the decisions a single rank would -- whatever the schedule of the plain products is."""
        self.step_launches = []
        if retain_sparsity or (filter_eps is not None and filter_eps > 0.0):
            works, staged = self._post_all()
            for w in works:
                w.wait()
            for item in staged:
                if isinstance(item, tuple):
                    item[1].copy_(item[0])
            Cout, counts = self.eng.multiply_local(alpha, self.A_panel, self.B_panel, beta, self.C_in, retain_sparsity=retain_sparsity,
                                                   filter_eps=filter_eps or 0.0)
            self._launched(self.eng, getattr(self.eng, "last_launch_flop", counts.flop))
            return Cout, counts
        if self.mode in ("colpipe", "colpipe2d", "tilepipe"):
            if self._cbounds is None or (self.mode == "tilepipe") != (self._rbounds is not None):
                raise ValueError("CannonMultiply: mode '%s' must be chosen at construction (the images are laid out for it)" % self.mode)
            return self._multiply_tilepipe(alpha, beta) if self.mode == "tilepipe" else self._multiply_colpipe(alpha, beta)
        if self.mode == "gather":
            return self._multiply_gather(alpha, beta)
        g, eng = self.grid, self.eng
        # C's structure for all ticks at once (pattern-only symbolic product of the full panels)
        row_p, counts0 = eng.symbolic(self.A_panel, self.B_panel, self.C_in, retain_sparsity=False)
        Cout = eng.init_c(beta, self.C_in, row_p, counts0, self.dtype)
        works, v, a_data, b_data = self._post(0, 0)
        flop = nprod = 0
        self.last_tick_flop = 0
        for tick in range(g.nvirt):
            for w in works:
                w.wait()
            cur = (v, a_data, b_data)
            if tick + 1 < g.nvirt:  # next tick's panels travel while this tick multiplies
                works, v, a_data, b_data = self._post(tick + 1, (tick + 1) & 1)
            cv, ca, cb = cur
            Ai, Bi = self.A_img[cv], self.B_img[cv]
            A = DbcsrMatrix(Ai.row_blk_size, Ai.col_blk_size, Ai.row_p, Ai.col_i, Ai.blk_p, ca, "A")
            B = DbcsrMatrix(Bi.row_blk_size, Bi.col_blk_size, Bi.row_p, Bi.col_i, Bi.blk_p, cb, "B")
            if Ai.nblks and Bi.nblks:
                te = self.last_engine = self._engine(("tick", tick))   # one plan per tick
                cnt = te.accumulate(alpha, A, B, Cout)
                flop += cnt.flop
                nprod += cnt.nproducts
                self._launched(te, cnt.flop)
        counts0.flop, counts0.nproducts = flop, nprod
        return Cout, counts0

    # ------------------------------------------------------------------
    def gather_global(self, Cloc):
        """(rows, cols, blocks) of the whole C on rank 0 (tests only): global block coordinates and dense blocks."""
        P, g = self.part, self.grid
        rs, cs, row_p, col_i, blk_p, data = Cloc.to_host()
        rows = np.repeat(np.arange(len(rs)), np.diff(row_p))
        grow, gcol = P.rows_of[g.myprow][rows], P.cols_of[g.mypcol][col_i]
        blocks = [data[blk_p[b]:blk_p[b] + int(rs[rows[b]]) * int(cs[col_i[b]])].copy() for b in range(len(col_i))]
        mine = (grow, gcol, blocks)
        if g.world == 1:
            return [mine]
        out = [None] * g.world if g.rank == 0 else None
        dist.gather_object(mine, out, dst=0)
        return out
