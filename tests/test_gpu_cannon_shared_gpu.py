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
                                        (2, "colpipe"), (4, "colpipe+dist"), (4, "colpipe+filter"), (8, "colpipe")])
def test_cannon_hip_engine_ranks_share_one_gpu(world, mode):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "run_dist_check.py"), "gloo", mode]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "-> OK" in r.stdout, r.stdout[-2000:]
