"""The RCCL panel exchange of include/dbcsr_amd_comm.h on the one GPU a test box has: a one-rank communicator
(librccl loaded on demand, ncclCommInitRank), a grouped send-to-self / receive-from-self of block data and of an
int32 index array on the communication stream, the sizes allgather, and the stream / event ordering against the
compute stream.  (Two ranks cannot share one device under RCCL; the multi-rank schedule is covered by the gloo
tests, tests/test_cannon_gloo.py, and by the distributed-input tests.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_native_comm_self_exchange_and_allgather():
    from dbcsr_amd.comm import NativeComm
    comm = NativeComm()
    assert comm.world == 1 and comm.rank == 0
    dev = torch.device("cuda", torch.cuda.current_device())
    src = torch.arange(100003, dtype=torch.float64, device=dev) * 0.5   # produced on the compute stream
    idx = torch.arange(777, dtype=torch.int32, device=dev) * 3
    dst = torch.zeros_like(src)
    idst = torch.zeros_like(idx)
    ev = comm.exchange([(src, 0), (idx, 0)], [(dst, 0), (idst, 0)])
    torch.cuda.current_stream().wait_event(ev)
    assert torch.equal(dst, src) and torch.equal(idst, idx)
    sizes = torch.tensor([src.numel(), idx.numel()], dtype=torch.int64, device=dev)
    allsz = torch.zeros(2 * comm.world, dtype=torch.int64, device=dev)
    torch.cuda.current_stream().wait_event(comm.allgather_bytes(sizes, allsz))
    assert allsz.tolist() == [100003, 777]
    # invalid peers are rejected before anything is posted
    with pytest.raises(RuntimeError):
        comm.exchange([(src, 5)], [])
    comm.close()


def test_cannon_transport_selftest_on_one_rank():
    """the bounded-wait ring exchange CannonMultiply runs before it trusts the native transport (here: a ring of one)"""
    from dbcsr_amd.cannon import CannonMultiply
    from dbcsr_amd.comm import NativeComm
    plan = CannonMultiply.__new__(CannonMultiply)
    plan.comm = NativeComm()
    try:
        assert plan._native_selftest(timeout_s=30.0) is None
    finally:
        plan.comm.close()
