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
        self.L = _lib.load_library()
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        ident = C.create_string_buffer(128)
        if rank == 0 and self.L.dbcsr_amd_comm_unique_id(ident) != 0:
            raise RuntimeError("dbcsr_amd_comm_unique_id failed (is librccl available?)")
        if world > 1:  # the host's job: hand rank 0's id to everybody (MPI_Bcast in a Fortran host)
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = C.create_string_buffer(box[0], 128)
        self.h = C.c_void_p()
        if self.L.dbcsr_amd_comm_create(C.byref(self.h), ident, world, rank) != 0:
            raise RuntimeError("dbcsr_amd_comm_create failed")
        self.rank, self.world = rank, world
        self.stream = torch.cuda.Stream()  # communication stream: transfers overlap the local multiply

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
