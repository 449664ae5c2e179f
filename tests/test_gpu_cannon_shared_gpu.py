"""The multi-rank Cannon driver on the HIP engine: N processes share the one GPU of the test box (gloo rendezvous,
host-staged debug transport -- RCCL does not allow two ranks on one device), every rank multiplies its tiles with the
real kernels, rank 0 compares the gathered C with the CPU oracle's global multiply (tools/run_dist_check.py).
Covers what the gloo CPU tests cannot: the HIP engine's symbolic / init_c / in-place accumulate path under N > 1,
with both schedules (also on eight ranks: the 4 x 2 grid of BASELINE's 8-GPU configurations, two A images per rank), and (``+dist``) distributed input whose blocks start in HBM on arbitrary ranks: packing, exchange and
sorting of make_images (cannon.redistribute) on the device."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,mode", [(2, "gather"), (2, "ticks"), (4, "ticks"), (4, "gather"), (4, "ticks+dist"), (2, "gather+dist"), (4, "gather+filter"), (2, "ticks+filter"), (8, "ticks"), (8, "gather+dist"),
                                        (2, "colpipe"), (4, "colpipe+dist"), (4, "colpipe+filter"), (8, "colpipe"),
                                        (4, "colpipe2d"), (8, "colpipe2d"), (4, "colpipe2d+dist"),   # round 6: the column-chunk pipeline on the 2-D grid
                                        (4, "tilepipe"), (8, "tilepipe"), (4, "tilepipe+dist"), (4, "tilepipe+filter")])   # ... and A's images in row chunks too
def test_cannon_hip_engine_ranks_share_one_gpu(world, mode):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "run_dist_check.py"), "gloo", mode]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "-> OK" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("mode", ["gather", "ticks", "colpipe", "colpipe2d", "tilepipe", "gather+dist"])
def test_cannon_native_transport_on_a_one_rank_communicator(mode):
    """every schedule with transport="native" on the one GPU of the box: librccl is loaded, a one-rank RCCL communicator is made, its
    self-test runs, the ranks' agreement on the transport goes through torch's communicator, and the schedule's exchange code runs with
    the native communicator in place (no peer to talk to: the posts are empty) -- the path a multi-GPU node takes first"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "run_dist_check.py"), "nccl", mode, "native"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "-> OK" in r.stdout and "transport=native" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("mode", ["gather", "colpipe"])
def test_cannon_auto_transport_falls_back_when_rccl_refuses_the_shared_device(mode):
    """two ranks on ONE device ask for transport="auto": RCCL cannot serve them (two ranks of a communicator on one GPU), every rank
    must notice, all must agree on torch.distributed, and nobody may hang -- what happens on a node where the native exchange cannot be
    set up on some rank"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "run_dist_check.py"), "gloo", mode, "auto"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "-> OK" in r.stdout and "transport=torch" in r.stdout, r.stdout[-2000:]
