"""The checker must not oversubscribe the host it runs on: a GPU box may show 256 CPUs behind a cgroup quota of 16, and OpenMP's default of
one thread per visible CPU then makes single oracle calls of a millisecond take tens of seconds (profiles/r05_soak_multiproc.txt: the
environment of round 4's two unexplained `F`).  oracle.py bounds the team by the CPUs the process may actually use."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_team_is_bounded_by_the_usable_cpus():
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import oracle as O\n"
            "print(O.max_threads(), O.usable_cpus())\n") % ROOT
    env = {k: v for k, v in os.environ.items() if k != "OMP_NUM_THREADS"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    team, usable = (int(x) for x in out.stdout.split()[-2:])
    assert 1 <= team <= usable <= (os.cpu_count() or 1)
    # an explicit setting stands
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, OMP_NUM_THREADS="3"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and int(out.stdout.split()[-2]) == 3, out.stdout + out.stderr[-500:]


def test_bench_counts_the_cgroup_quota():
    sys.path.insert(0, ROOT)
    import bench
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            assert n <= max(1, int(int(q) / int(per) + 0.5))
    except OSError:
        pass
