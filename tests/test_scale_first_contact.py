"""CPU-side dry run of the first contact with a multi-GPU node (tools/scale_first_contact.sh): the stages the script
would run, the all-pairs exchange self-check on real processes (gloo, torch transport: the same patterns and checks the RCCL run makes),
and the exchange probe behind bench.py's `comm` object (bytes per peer, time of the exchange alone) under every grid the tests use."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_dry_run_lists_the_stages_in_order():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_first_contact.sh"), "8", "--dry-run"], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    stages = [ln for ln in r.stdout.splitlines() if ln.startswith("== stage ")]
    assert [s.split()[2].rstrip(":") for s in stages] == ["1_comm_selfcheck", "2_dist_check_gather", "2_dist_check_ticks", "2_dist_check_colpipe",
                                                           "2_dist_check_gather_dist", "3_bench"]
    assert "comm_selfcheck.py nccl native" in stages[0] and "--nproc-per-node 8" in stages[0]
    assert all("run_dist_check.py nccl" in s and s.rstrip().endswith("native") for s in stages[1:5])
    assert "bench.py --gpus 8" in stages[5] and "--dist-transport native" in stages[5]


@pytest.mark.parametrize("world", [2, 3])
def test_comm_selfcheck_on_cpu_processes(world):
    env = dict(os.environ, COMM_SELFCHECK_CPU="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "comm_selfcheck.py"), "gloo", "torch"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["comm_selfcheck"] == "OK" and d["ranks"] == world and d["pairs_checked"] == world * (world - 1) and d["rccl_ranks"] == 0


def _probe_worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dbcsr_amd import cannon
        from tests.cpu_backend import OracleBackend
        plan = cannon.CannonMultiply(23 * 9 + 16, 13 * 14 + 5, 17 * 8 + 3, (0.6, 0.65, 0.8), [1, 23], dtype=torch.float64, engine=OracleBackend(),
                                     device=torch.device("cpu"), mix_n=[1, 13], mix_k=[2, 17, 1, 5], mode=mode)
        plan.multiply(1.0, 1.0)
        cp = plan.comm_probe()
        g, r, c = plan.grid, plan.grid.myprow, plan.grid.mypcol
        want = sum(8 * plan.A_img[v].data_numel for v in range(g.nvirt) if g.a_owner(r, v) != g.rank) + \
            sum(8 * plan.B_img[v].data_numel for v in range(g.nvirt) if g.b_owner(v, c) != g.rank)
        peers = len({g.a_owner(r, v) for v in range(g.nvirt) if g.a_owner(r, v) != g.rank and plan.A_img[v].data_numel} |
                    {g.b_owner(v, c) for v in range(g.nvirt) if g.b_owner(v, c) != g.rank and plan.B_img[v].data_numel})
        ok = cp["bytes_in"] == want and cp["peers_in"] == peers and 0 < cp["max_bytes_from_one_peer"] <= cp["bytes_in"] and cp["ms"] > 0
        # a multiply after the probe still gives the same result: the probe moved the images into the places a step puts them
        t = torch.tensor([1 if ok else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            q.put((int(t.item()), g.nprows, g.npcols))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "gather"), (4, "gather"), (6, "ticks"), (4, "colpipe")])
def test_comm_probe_counts_what_a_step_moves(world, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_probe_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, pr, pc = q.get(timeout=240)
    for p in procs:
        p.join(60)
    assert ok == 1 and pr * pc == world
