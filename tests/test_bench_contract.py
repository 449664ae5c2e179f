"""bench.py's side of the driver contract that can be checked without a GPU: its workloads are BASELINE.json's configurations, the
default run is configs[1] on one GPU, `--gpus N` from a bare shell re-launches itself under torch.distributed.run with one rank per
GPU on 127.0.0.1, and the command line is handed on unchanged."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_workloads_are_the_baseline_configurations(monkeypatch):
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert len(base["configs"]) == 5
    expect = {"config1": (4096, 0.10, [1, 4], "f64"), "config2": (32768, 0.10, [1, 23], "f64"), "config3": (32768, 0.05, [1, 13, 1, 23, 1, 32], "f64"),
              "config4": (131072, 0.01, [1, 23], "f64"), "config5": (131072, 0.20, [1, 32], "f32")}
    for key, (m, fill, mix, dt) in expect.items():
        names = [n for n in bench.WORKLOADS if n.startswith(key + "_")]
        assert len(names) == 1, key
        M, N, K, f, mx, d = bench.WORKLOADS[names[0]]
        assert (M, N, K, f, mx, d) == (m, m, m, fill, mix, dt)
        assert str(m) in base["configs"][int(key[-1]) - 1].replace("×", "x")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert a.gpus == 1 and a.workload.startswith("config2_") and a.steps > 0 and a.warmup >= 0


def test_gpus_n_launches_one_rank_per_gpu(monkeypatch):
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(bench.subprocess if hasattr(bench, "subprocess") else __import__("subprocess"), "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    args = bench.parse_args()
    assert bench.self_launch(args) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_n_rank_roofline_is_the_sum_over_a_steps_launches():
    """for N > 1 `roofline.kernel_ms` is the SUM over the rank's block-product launches of a step with `flop_per_launch`
    to match, so that `frac` compares with the one-GPU line (whose single launch is the whole multiply)."""
    import bench
    launches = [(1.25, 40 * 10 ** 9), (1.5, 50 * 10 ** 9), (1.25, 42 * 10 ** 9)]
    r = bench.step_roofline(launches, "f64", "mm_numeric_f64_hot<23,23,23>")
    assert r["launches_per_step"] == 3 and abs(r["kernel_ms"] - 4.0) < 1e-12 and r["flop_per_launch"] == 132 * 10 ** 9
    assert abs(r["achieved"] - 132e9 / 4.0e-3 / 1e12) < 1e-9 and abs(r["frac"] - r["achieved"] / bench.FP64_MFMA_PEAK_TFLOPS) < 1e-12
    assert r["unit"] == "TFLOP/s" and r["bound"] == "mfma" and r["traffic"] is None
    assert bench.step_roofline(launches, "f32", "k")["peak"] == 157.3


def test_bench_line_fields_of_round_5(monkeypatch):
    """the fields the round added are spelled as DESIGN 5 says: the budget switch of the other configurations' counter passes, thread teams
    sized by the CPUs the process may use"""
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--other-pmc-budget", "60"])
    a = bench.parse_args()
    assert a.other_pmc_budget == 60.0
    assert 1 <= bench.usable_cpus() <= (os.cpu_count() or 1)
    src = open(os.path.join(ROOT, "bench.py")).read()
    for field in ('"traffic"', '"launches_per_step"', '"kernel_ms_note"', '"traffic_note"', '"cores"', '"ms_per_step_cold"', '"other_configs"'):
        assert field in src, field
