"""The exact-size stack kernel of the acc ABI (csrc/smm_exact.h, compiled per (m, n, k) at run time the first time a homogeneous stack of
at least 256 entries of the triplet arrives -- the reference compiles its kernels the same way, src/acc/libsmm_acc/libsmm_acc.cpp:90-195,
281-321) against the CPU oracle: the reference's validation inputs (integer values, EXACT equality, libsmm_acc.cpp:55-87) and random values
at the north star's tolerance; B transposed by libsmm_acc_transpose and as stored; runs of equal C offsets that straddle waves; unsorted
stacks; a stack whose length is not a multiple of the group."""
import numpy as np
import pytest

from dbcsr_amd import lib as L
from oracle import oracle as O
from tests.gpu_util import rel_err, run_stack

pytestmark = pytest.mark.gpu

# cubes of the reference's table, the mixes of config 3, every remainder of k modulo 4, leading dimensions of 16 and 32
# (padded LDS pitch), single-tile and four-tile blocks
TRIPLETS = [(23, 23, 23), (13, 13, 13), (32, 32, 32), (4, 4, 4), (5, 5, 5), (13, 23, 32), (32, 13, 23), (23, 32, 13), (16, 16, 16),
            (32, 16, 32), (16, 32, 16), (9, 8, 5), (24, 24, 24), (1, 1, 1), (7, 1, 31), (29, 30, 31), (25, 26, 27), (8, 9, 18), (32, 32, 1),
            (1, 32, 32), (17, 3, 22)]


def last_kernel():
    return L.load_library().dbcsr_amd_smm_last_kernel().decode()


def expect_exact(m, n, k):
    """automatic mode: triplets from 8^3 on (below, a stack is one launch-bound kernel whatever its body: the ahead-of-time kernels stay)"""
    return m * n * k >= 512


@pytest.mark.parametrize("bt", [True, False])
@pytest.mark.parametrize("m,n,k", TRIPLETS)
def test_exact_kernel_integer_inputs_equal_the_oracle(m, n, k, bt):
    na, nb, nc, nstack = 100, 100, 30, 1001   # 1001: the last wave's group is short
    a = O.mat_init(na, m, k, 42)
    b = O.mat_init(nb, k, n, 24)
    stack = O.stack_init(nstack, nc, na, nb, m, n, k, rseed=7)
    c_ref = np.zeros(nc * m * n)
    O.stack_calc(stack, c_ref, a, b, m, n, k, b_transposed=False)
    # (max_kernel_dim below the block: the host's rule says B stays as stored, libsmm_acc.cpp:485)
    rc, c = run_stack(stack, a, b, np.zeros(nc * m * n), m, n, k, L.dbcsr_type_real_8, max_kernel_dim=80 if bt else 0, transpose_b=bt)
    assert rc >= 0
    assert last_kernel().startswith(("smm_stack_f64_exact<%d,%d,%d" if expect_exact(m, n, k) else "smm_stack_f64_lds(%d,%d,%d") % (m, n, k)), last_kernel()
    assert ("transposed" in last_kernel()) == bt
    assert np.array_equal(c, c_ref)


@pytest.mark.parametrize("m,n,k", [(23, 23, 23), (13, 23, 32), (32, 32, 32), (5, 5, 5)])
def test_exact_kernel_random_values_sorted_and_unsorted(m, n, k):
    rng = np.random.default_rng(m + 100 * n + 10000 * k)
    na, nb, nc, nstack = 300, 300, 40, 3000
    a, b, c0 = rng.random(na * m * k), rng.random(nb * k * n), rng.random(nc * m * n)
    for sort in (True, False):
        stack = np.empty(3 * nstack, np.int32)
        ci = rng.integers(0, nc, nstack)
        stack[0::3] = rng.integers(0, na, nstack) * m * k + 1
        stack[1::3] = rng.integers(0, nb, nstack) * k * n + 1
        stack[2::3] = (np.sort(ci) if sort else ci) * m * n + 1
        c_ref = c0.copy()
        O.stack_calc(stack, c_ref, a, b, m, n, k, b_transposed=False)
        rc, c = run_stack(stack, a, b, c0.copy(), m, n, k, L.dbcsr_type_real_8)
        assert rc >= 0
        assert last_kernel().startswith("smm_stack_f64_exact<" if expect_exact(m, n, k) else "smm_stack_f64_lds(")
        assert rel_err(c, c_ref) <= 1e-10


def test_one_long_run_straddles_many_waves():
    # every entry adds to the same C block: 2000 entries = 125 waves whose atomics meet in one block
    m, n, k = 23, 23, 23
    rng = np.random.default_rng(5)
    na, nb, nstack = 50, 50, 2000
    a, b = rng.random(na * m * k), rng.random(nb * k * n)
    stack = np.empty(3 * nstack, np.int32)
    stack[0::3] = rng.integers(0, na, nstack) * m * k + 1
    stack[1::3] = rng.integers(0, nb, nstack) * k * n + 1
    stack[2::3] = 1
    c_ref = np.zeros(m * n)
    O.stack_calc(stack, c_ref, a, b, m, n, k, b_transposed=False)
    rc, c = run_stack(stack, a, b, np.zeros(m * n), m, n, k, L.dbcsr_type_real_8)
    assert rc >= 0 and last_kernel().startswith("smm_stack_f64_exact<23,23,23")
    assert rel_err(c, c_ref) <= 1e-10


def test_short_stacks_of_an_unseen_triplet_do_not_compile():
    # 100 entries of a triplet nobody asked for before: the run-time-size kernel takes it (no 0.4 s compilation for 100 products)
    m, n, k = 11, 12, 10
    na, nb, nc, nstack = 20, 20, 5, 100
    a = O.mat_init(na, m, k, 42)
    b = O.mat_init(nb, k, n, 24)
    stack = O.stack_init(nstack, nc, na, nb, m, n, k, rseed=7)
    c_ref = np.zeros(nc * m * n)
    O.stack_calc(stack, c_ref, a, b, m, n, k, b_transposed=False)
    rc, c = run_stack(stack, a, b, np.zeros(nc * m * n), m, n, k, L.dbcsr_type_real_8)
    assert rc >= 0 and last_kernel().startswith("smm_stack_f64_lds(11,12,10")
    assert np.array_equal(c, c_ref)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DBCSR_AMD_SWEEP_EXACT_STACKS", "24"))))
def test_random_triplets_through_the_exact_kernel(seed):
    """random (m, n, k) up to 32 (every one compiles its kernel: ~0.4 s), stacks of 256 ... 3000 entries sorted, binned or shuffled, B transposed or
    as stored: integer-valued inputs, so the result must be EXACT whatever the summation order"""
    rng = np.random.default_rng(4200 + seed + int(__import__("os").environ.get("DBCSR_AMD_SWEEP_OFFSET", "0")))
    m, n, k = (int(x) for x in rng.integers(1, 33, size=3))
    while m * n * k < 512:
        m, n, k = (int(x) for x in rng.integers(4, 33, size=3))
    nstack = int(rng.choice([256, 257, 271, 1000, 1023, 3000]))
    na, nb = int(rng.integers(1, 80)), int(rng.integers(1, 80))
    nc = int(rng.integers(1, max(2, nstack // 8)))
    a = O.mat_init(na, m, k, 42)
    b = O.mat_init(nb, k, n, 24)
    stack = O.stack_init(nstack, nc, na, nb, m, n, k, rseed=int(rng.integers(1, 1000)))
    order = seed % 3
    if order:   # 1: shuffled entries (runs of length one), 2: binned by c offset as the host does for small blocks
        ent = np.asarray(stack, np.int32).reshape(-1, 3)
        ent = ent[rng.permutation(len(ent))] if order == 1 else ent[np.argsort((ent[:, 2].astype(np.int64) * (ent[:, 2] + 3)) % 4096, kind="stable")]
        stack = np.ascontiguousarray(ent.reshape(-1))
    bt = bool(seed % 2)
    c_ref = np.zeros(nc * m * n)
    O.stack_calc(stack, c_ref, a, b, m, n, k, b_transposed=False)
    rc, c = run_stack(stack, a, b, np.zeros(nc * m * n), m, n, k, L.dbcsr_type_real_8, max_kernel_dim=80 if bt else 0, transpose_b=bt)
    assert rc >= 0
    assert last_kernel().startswith("smm_stack_f64_exact<%d,%d,%d" % (m, n, k)), last_kernel()
    assert np.array_equal(c, c_ref), (m, n, k, nstack, order, bt)


def test_host_threads_compile_and_run_different_triplets_at_once():
    """what the real host does on its first multiply (src/mm/dbcsr_mm_accdrv.F: one stream per OpenMP thread, core/dbcsr_lib.F:248-262): several host threads
    meet triplets nobody compiled yet AT THE SAME TIME -- the run-time compilation is serialised inside the library, every thread gets its kernel, every stack
    its exact result"""
    import threading
    triplets = [(9 + 2 * t, 31 - 3 * t, 12 + t) for t in range(6)] + [(21, 21, 21), (21, 21, 21)]   # (two threads share a triplet)
    results, errors = {}, []
    start = threading.Barrier(len(triplets))

    def worker(t):
        try:
            m, n, k = triplets[t]
            na, nb, nc, nstack = 40, 50, 12, 700 + t
            a = O.mat_init(na, m, k, 42 + t)
            b = O.mat_init(nb, k, n, 24 + t)
            stack = O.stack_init(nstack, nc, na, nb, m, n, k, rseed=7 + t)
            c_ref = np.zeros(nc * m * n)
            O.stack_calc(stack, c_ref, a, b, m, n, k, b_transposed=False)
            start.wait()
            rc, c = run_stack(stack, a, b, np.zeros(nc * m * n), m, n, k, L.dbcsr_type_real_8)
            results[t] = (rc, np.array_equal(c, c_ref), last_kernel())
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
            start.abort()

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(len(triplets))]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=300)
    assert not errors, errors
    for t, (m, n, k) in enumerate(triplets):
        rc, same, name = results[t]
        assert rc >= 0 and same, (t, m, n, k, name)
        assert name.startswith("smm_stack_f64_exact<%d,%d,%d" % (m, n, k)), name
