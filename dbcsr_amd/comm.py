"""RCCL panel exchange through the C-ABI (include/dbcsr_amd_comm.h): what an MPI / Fortran host would call for the
Cannon shift (reference: device-pointer isend / irecv in multiply_cannon_g2g, src/mm/dbcsr_mm_cannon.F:2528-2557).
Here the 128-byte communicator id travels over torch.distributed (any backend) instead of MPI_Bcast; after that no
torch collective is involved: sends and receives of device tensors are posted as ONE RCCL group on a dedicated HIP
stream, and the compute stream waits on an event."""
import ctypes as C

import torch
import torch.distributed as dist

from . import lib as _lib
from .matrix import StreamHandle


class NativeComm:
    def __init__(self, group=None):
        """Collective over the group.  Every step that can fail locally is AGREED ON before the next collective step, so that a
        rank without a usable librccl makes all ranks raise together instead of leaving the others inside a broadcast or inside
        ncclCommInitRank (the caller falls back to torch.distributed)."""
        self.h = None
        self.L = _lib.load_library()
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rank, self.world = rank, world
        ident = C.create_string_buffer(128)
        ok = 1 if self.L.dbcsr_amd_comm_available() == 1 else 0
        if ok and rank == 0 and self.L.dbcsr_amd_comm_unique_id(ident) != 0:
            ok = 0
        if not self._agree(ok, group):
            raise RuntimeError("librccl not usable on some rank (dbcsr_amd_comm_available / dbcsr_amd_comm_unique_id)")
        if world > 1:  # the host's job: hand rank 0's id to everybody (MPI_Bcast in a Fortran host)
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        ok = 1 if self.L.dbcsr_amd_comm_create(C.byref(h), ident, world, rank) == 0 else 0
        if ok:
            self.h = h
        if not self._agree(ok, group):
            self.close()
            raise RuntimeError("dbcsr_amd_comm_create failed on some rank")
        self.stream = torch.cuda.Stream()  # communication stream: transfers overlap the local multiply

    @staticmethod
    def _agree(ok, group):
        """logical AND of a local flag over the group, through torch's own communicator"""
        if not (dist.is_initialized() and dist.get_world_size(group) > 1):
            return bool(ok)
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor([int(ok)], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return int(t.item()) == 1

    def close(self):
        if getattr(self, "h", None):
            self.L.dbcsr_amd_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def exchange(self, sends, recvs):
        """sends / recvs: lists of (device tensor, peer).  Posts one RCCL group on the communication stream, after
        everything already queued on the current (compute) stream; returns an event the consumer stream waits on."""
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self.stream.wait_event(ready)  # the buffers being sent were produced on the compute stream

        def ops(lst):
            arr = (_lib.CommOp * max(1, len(lst)))()
            for i, (t, peer) in enumerate(lst):
                arr[i] = _lib.CommOp(t.data_ptr(), t.numel() * t.element_size(), int(peer), 0)
            return arr

        s, r = ops(sends), ops(recvs)
        rc = self.L.dbcsr_amd_comm_exchange(self.h, s, len(sends), r, len(recvs), StreamHandle(self.stream).ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_comm_exchange failed (%d)" % rc)
        done = torch.cuda.Event()
        done.record(self.stream)
        return done

    def allgather_bytes(self, t_send, t_recv):
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self.stream.wait_event(ready)  # t_send was produced on the compute stream
        rc = self.L.dbcsr_amd_comm_allgather(self.h, t_send.data_ptr(), t_recv.data_ptr(), t_send.numel() * t_send.element_size(),
                                             StreamHandle(self.stream).ptr)
        if rc != 0:
            raise RuntimeError("dbcsr_amd_comm_allgather failed (%d)" % rc)
        done = torch.cuda.Event()
        done.record(self.stream)
        return done
