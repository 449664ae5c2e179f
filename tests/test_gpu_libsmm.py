"""GPU parity of the libsmm_acc C-ABI (libsmm_acc_transpose + libsmm_acc_process +
c_calculate_norms) against the CPU oracle.  Mirrors the reference's kernel
validation (src/acc/libsmm_acc/libsmm_acc.cpp:55-87: integer-valued inputs,
EXACT checksum equality) and acc_bench's tolerance test on random inputs."""
import numpy as np
import pytest
import torch

from dbcsr_amd import lib as L
from oracle import oracle as O
from tests.gpu_util import StreamHandle, rel_err, run_stack

pytestmark = pytest.mark.gpu

# tuned flagship triplets of the reference + GPU block mixes of tests/dbcsr_unittest3.F:79-120 + odd shapes
TRIPLETS = [(23, 23, 23), (13, 13, 13), (32, 32, 32), (4, 4, 4), (5, 7, 9), (1, 3, 4), (13, 23, 32), (32, 13, 23), (23, 16, 23),
            (16, 23, 16), (14, 29, 32), (4, 13, 25), (9, 8, 5), (24, 24, 24), (45, 67, 78), (78, 45, 67), (1, 1, 1), (7, 1, 33),
            # round 5: the workgroup-per-entry-group kernel of the blocks of 33 ... 80 (smm_stack_f64_big): every sub-block shape edge, inner
            # dimensions with every remainder modulo 4 and 16
            (72, 72, 72), (80, 80, 80), (64, 64, 64), (40, 40, 40), (33, 33, 33), (55, 55, 55), (23, 23, 78), (80, 16, 37), (13, 72, 33),
            (48, 56, 17), (48, 56, 18), (48, 56, 19), (41, 49, 35), (80, 80, 1), (33, 80, 64),
            # round 6: sub-blocks in units of 4 x 4 (BigSub): odd / even halves in either dimension
            (34, 34, 34), (35, 36, 37), (37, 33, 36), (53, 64, 41), (65, 73, 16), (69, 77, 31), (36, 36, 36), (44, 52, 20), (57, 61, 9)]


@pytest.mark.parametrize("m,n,k", TRIPLETS)
def test_validate_kernel_exact(m, n, k):
    # the reference's own validation inputs: n_a = n_b = 100, n_c = 10, stack 100 (libsmm_acc_benchmark.cpp:45-52)
    na, nb, nc, nstack = 100, 100, 10, 100
    a = O.mat_init(na, m, k, 42)
    b = O.mat_init(nb, k, n, 24)
    stack = O.stack_init(nstack, nc, na, nb, m, n, k, rseed=7)
    c_ref = np.zeros(nc * m * n)
    O.stack_calc(stack, c_ref, a, b, m, n, k, b_transposed=False)
    rc, c = run_stack(stack, a, b, np.zeros(nc * m * n), m, n, k, L.dbcsr_type_real_8)
    assert rc >= 0
    assert O.lib().orc_check_sum(c, c.size) == O.lib().orc_check_sum(c_ref, c_ref.size)  # exact, as validate_kernel
    assert np.array_equal(c, c_ref)


def test_tuning_stack_23_random_values():
    # tuner/timer configuration (libsmm_acc_benchmark.cpp:36-44) with U(0,1) values, stack sorted by c as accdrv does
    m = n = k = 23
    na, nb, nc, nstack = 10000, 10000, 1000, 16005
    rng = np.random.default_rng(1)
    a, b, c0 = rng.random(na * m * k), rng.random(nb * k * n), rng.random(nc * m * n)
    stack = O.stack_init(nstack, nc, na, nb, m, n, k, rseed=3)
    c_ref = c0.copy()
    O.stack_calc(stack, c_ref, a, b, m, n, k, b_transposed=False)
    rc, c = run_stack(stack, a, b, c0.copy(), m, n, k, L.dbcsr_type_real_8)
    assert rc >= 0
    assert rel_err(c, c_ref) <= 1e-10  # north-star tolerance on C values


def test_unsorted_stack_and_large_blocks():
    # c offsets in random order (runs of length 1) and a block above max_kernel_dim (B not transposed: libsmm_acc.cpp:485)
    rng = np.random.default_rng(2)
    for (m, n, k) in [(23, 23, 23), (100, 90, 85)]:
        na, nb, nc, nstack = 50, 60, 7, 333
        a, b, c0 = rng.random(na * m * k), rng.random(nb * k * n), rng.random(nc * m * n)
        stack = np.empty(3 * nstack, np.int32)
        stack[0::3] = rng.integers(0, na, nstack) * m * k + 1
        stack[1::3] = rng.integers(0, nb, nstack) * k * n + 1
        stack[2::3] = rng.integers(0, nc, nstack) * m * n + 1
        c_ref = c0.copy()
        O.stack_calc(stack, c_ref, a, b, m, n, k, b_transposed=False)
        rc, c = run_stack(stack, a, b, c0.copy(), m, n, k, L.dbcsr_type_real_8)
        assert rc >= 0
        assert rel_err(c, c_ref) <= 1e-10


def test_fp32_stack():
    # the reference returns -10 for fp32 (libsmm_acc.cpp:338); this library runs it on the device
    m = n = k = 32
    na, nb, nc, nstack = 200, 200, 20, 2000
    rng = np.random.default_rng(3)
    a = rng.random(na * m * k, dtype=np.float32)
    b = rng.random(nb * k * n, dtype=np.float32)
    c0 = rng.random(nc * m * n, dtype=np.float32)
    stack = O.stack_init(nstack, nc, na, nb, m, n, k, rseed=5)
    c_ref = c0.astype(np.float64)
    O.stack_calc(stack, c_ref, a.astype(np.float64), b.astype(np.float64), m, n, k, b_transposed=False)
    rc, c = run_stack(stack, a, b, c0.copy(), m, n, k, L.dbcsr_type_real_4, transpose_b=True)  # transpose is a no-op for fp32
    assert rc >= 0
    assert rel_err(c, c_ref) <= 2e-3 * 1e-1  # acc_bench's fp32 epsilon is 2e-3 (acc_bench.c:69-71); we are far inside


@pytest.mark.parametrize("m,n,k", [(13, 23, 7), (5, 9, 31), (32, 1, 3), (40, 33, 35)])
def test_fp32_stack_odd_sizes(m, n, k):
    na, nb, nc, nstack = 60, 70, 9, 500
    rng = np.random.default_rng(m * 1000 + n * 10 + k)
    a = rng.random(na * m * k, dtype=np.float32)
    b = rng.random(nb * k * n, dtype=np.float32)
    c0 = rng.random(nc * m * n, dtype=np.float32)
    stack = O.stack_init(nstack, nc, na, nb, m, n, k, rseed=11)
    c_ref = c0.astype(np.float64)
    O.stack_calc(stack, c_ref, a.astype(np.float64), b.astype(np.float64), m, n, k, b_transposed=False)
    rc, c = run_stack(stack, a, b, c0.copy(), m, n, k, L.dbcsr_type_real_4)
    assert rc >= 0
    assert float(np.max(np.abs(c - c_ref))) <= 2e-5 * float(np.max(np.abs(c_ref)))


@pytest.mark.parametrize("seed,nstack", [(11, 700), (12, 1), (13, 33), (14, 2500)])
def test_inhomogeneous_stack_runs_on_the_device(seed, nstack):
    """A stack with different (m, n, k) per entry (def_mnk = 0): the reference's library refuses it (libsmm_acc.cpp:324-339 returns
    -1, the host multiplies it on the CPU and aborts in G2G mode); here it is read from the host's own 7-integer records (unsorted,
    ordinary host memory) and run on the device.  B blocks with both dims <= max_kernel_dim arrive transposed, the others as stored."""
    rng = np.random.default_rng(seed)
    lib = L.load_library()
    st = StreamHandle()
    sizes = [4, 13, 23, 32, 7, 45, 90]
    max_dim = 80
    # data areas: blocks of random sizes back to back
    ent, a_parts, b_parts, c_off, c_sizes = [], [], [], [], []
    a_len = b_len = 0
    c_blocks = [(int(rng.choice(sizes)), int(rng.choice(sizes))) for _ in range(40)]
    c_pos = np.concatenate([[0], np.cumsum([m * n for m, n in c_blocks])])
    for _ in range(nstack):
        ci = int(rng.integers(0, len(c_blocks)))
        m, n = c_blocks[ci]
        k = int(rng.choice(sizes))
        A = rng.random((m, k))
        B = rng.random((k, n))
        a_parts.append(A.flatten(order="F"))
        bt = k <= max_dim and n <= max_dim
        b_parts.append((B.T if bt else B).flatten(order="F"))   # what libsmm_acc_transpose leaves behind: n x k column-major
        ent.append((m, n, k, a_len + 1, b_len + 1, int(c_pos[ci]) + 1, ci + 1, A, B))
        a_len += m * k
        b_len += k * n
    a = np.concatenate(a_parts)
    b = np.concatenate(b_parts)
    c0 = rng.random(int(c_pos[-1]))
    c_ref = c0.copy()
    for (m, n, k, ao, bo, co, ci, A, B) in ent:
        blk = c_ref[co - 1:co - 1 + m * n].reshape((m, n), order="F")
        blk += A @ B
    host = np.ascontiguousarray(np.array([e[:7] for e in ent], np.int32))   # 7 x nstack, column-major == rows of 7
    ta, tb, tc = torch.as_tensor(a).cuda(), torch.as_tensor(b).cuda(), torch.as_tensor(c0.copy()).cuda()
    dummy = torch.zeros(3 * nstack, dtype=torch.int32, device="cuda")
    rc = lib.libsmm_acc_process(host.ctypes.data, dummy.data_ptr(), nstack, L.dbcsr_type_real_8, ta.data_ptr(), tb.data_ptr(), tc.data_ptr(),
                                max(sizes), max(sizes), max(sizes), max_dim, 0, st.ptr, st.ptr)
    assert rc == 0
    host[:] = -12345   # the host reuses its records right after the call
    torch.cuda.synchronize()
    assert rel_err(tc.cpu().numpy(), c_ref) <= 1e-10


def test_return_codes():
    lib = L.load_library()
    st = StreamHandle()
    z = torch.zeros(64, dtype=torch.float64, device="cuda")
    s = torch.ones(3, dtype=torch.int32, device="cuda")
    # inhomogeneous stack without the host's records (or not fp64) -> -1, complex -> -10 : "run this stack on the CPU", never abort
    assert lib.libsmm_acc_process(None, s.data_ptr(), 1, L.dbcsr_type_real_8, z.data_ptr(), z.data_ptr(), z.data_ptr(), 4, 4, 4, 80, 0,
                                  st.ptr, st.ptr) == -1
    h = np.array([4, 4, 4, 1, 1, 1, 1], np.int32)
    assert lib.libsmm_acc_process(h.ctypes.data, s.data_ptr(), 1, L.dbcsr_type_real_4, z.data_ptr(), z.data_ptr(), z.data_ptr(), 4, 4, 4, 80, 0,
                                  st.ptr, st.ptr) == -1
    assert lib.libsmm_acc_process(None, s.data_ptr(), 1, L.dbcsr_type_complex_8, z.data_ptr(), z.data_ptr(), z.data_ptr(), 2, 2, 2, 80,
                                  1, st.ptr, st.ptr) == -10
    assert lib.libsmm_acc_process(None, s.data_ptr(), 0, L.dbcsr_type_real_8, z.data_ptr(), z.data_ptr(), z.data_ptr(), 4, 4, 4, 80, 1,
                                  st.ptr, st.ptr) == 0  # empty stack
    torch.cuda.synchronize()


@pytest.mark.parametrize("m,n", [(23, 23), (5, 9), (32, 13), (1, 7), (80, 80)])
def test_transpose_matches_oracle(m, n):
    lib = L.load_library()
    st = StreamHandle()
    nblk = 37
    data = O.mat_init(nblk, m, n, 42)
    trs = (np.arange(nblk, dtype=np.int32)[::-1] * m * n).copy()  # any order, 0-based offsets
    ref = data.copy()
    O.transpose(trs, ref, m, n)
    d = torch.as_tensor(data).cuda()
    t = torch.as_tensor(np.concatenate([[-1, -1], trs]).astype(np.int32)).cuda()  # exercise the `offset` argument
    assert lib.libsmm_acc_transpose(t.data_ptr(), 2, nblk, d.data_ptr(), L.dbcsr_type_real_8, m, n, 80, st.ptr) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), ref)


def test_norms_match_oracle():
    lib = L.load_library()
    st = StreamHandle()
    rng = np.random.default_rng(4)
    sizes = rng.integers(1, 700, 300).astype(np.int32)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    mat = rng.standard_normal(int(sizes.sum()))
    ref = O.norms(mat, offs, sizes)
    dm, do, dn = torch.as_tensor(mat).cuda(), torch.as_tensor(offs).cuda(), torch.as_tensor(sizes).cuda()
    out = torch.zeros(len(sizes), dtype=torch.float32, device="cuda")
    assert lib.c_calculate_norms(dm.data_ptr(), len(sizes), do.data_ptr(), dn.data_ptr(), out.data_ptr(), st.ptr) == 0
    torch.cuda.synchronize()
    assert np.allclose(out.cpu().numpy(), ref, rtol=1e-6)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DBCSR_AMD_SWEEP_STACKS", "0"))))
def test_random_triplets_and_stacks_exact(seed):
    """(off by default: set DBCSR_AMD_SWEEP_STACKS=N; 120 seeds run on MI355X in round 3, tools/gpu_sessions/r03_27_stack_sweeps.sh.)  libsmm_acc_process on random (m, n, k) up to 45 (beyond
    32: the direct kernel), random stack lengths around the group sizes and c offsets sorted, binned or shuffled: integer-valued inputs, so
    the result must be EXACT whatever the summation order."""
    rng = np.random.default_rng(900 + seed + int(__import__("os").environ.get("DBCSR_AMD_SWEEP_OFFSET", "0")))
    m, n, k = (int(x) for x in rng.integers(1, 46, size=3))
    if seed % 4 == 0:   # the LDS-staged range
        m, n, k = (int(x) for x in rng.integers(1, 33, size=3))
    nstack = int(rng.choice([2, 7, 8, 9, 15, 16, 17, 31, 33, 100, 257]))
    na, nb = int(rng.integers(1, 60)), int(rng.integers(1, 60))
    nc = int(rng.integers(1, max(2, nstack // 2 + 1)))   # INIT_STACK needs at least two entries per C block on average
    a = O.mat_init(na, m, k, 42)
    b = O.mat_init(nb, k, n, 24)
    stack = O.stack_init(nstack, nc, na, nb, m, n, k, rseed=int(rng.integers(1, 1000)))
    order = seed % 3
    if order:   # 1: shuffled entries (runs of length one), 2: binned by c offset as the host does for small blocks
        ent = np.asarray(stack, np.int32).reshape(-1, 3)
        ent = ent[rng.permutation(len(ent))] if order == 1 else ent[np.argsort((ent[:, 2].astype(np.int64) * (ent[:, 2] + 3)) % 4096, kind="stable")]
        stack = np.ascontiguousarray(ent.reshape(-1))
    c_ref = np.zeros(nc * m * n)
    O.stack_calc(stack, c_ref, a, b, m, n, k, b_transposed=False)
    rc, c = run_stack(stack, a, b, np.zeros(nc * m * n), m, n, k, L.dbcsr_type_real_8)
    assert rc >= 0
    assert np.array_equal(c, c_ref), (m, n, k, nstack, order)


@pytest.mark.parametrize("m,n,k", [(72, 72, 72), (45, 67, 78), (40, 33, 35), (23, 23, 50)])
def test_large_block_stack_with_b_as_stored_and_random_order(m, n, k):
    """smm_stack_f64_big with B NOT transposed (a host whose max_kernel_dim is below the block size skips libsmm_acc_transpose,
    libsmm_acc.cpp:485) and with C offsets in random order (runs of length one: a flush per entry), random values"""
    rng = np.random.default_rng(11)
    na, nb, nc, nstack = 40, 50, 6, 301
    a, b, c0 = rng.random(na * m * k), rng.random(nb * k * n), rng.random(nc * m * n)
    stack = np.empty(3 * nstack, np.int32)
    stack[0::3] = rng.integers(0, na, nstack) * m * k + 1
    stack[1::3] = rng.integers(0, nb, nstack) * k * n + 1
    stack[2::3] = rng.integers(0, nc, nstack) * m * n + 1
    c_ref = c0.copy()
    O.stack_calc(stack, c_ref, a, b, m, n, k, b_transposed=False)
    # max_kernel_dim below the block's dimensions: no transposition, libsmm_acc_process is told so through the same argument
    rc, c = run_stack(stack, a, b, c0.copy(), m, n, k, L.dbcsr_type_real_8, max_kernel_dim=20, transpose_b=True)
    assert rc >= 0
    assert rel_err(c, c_ref) <= 1e-10
    # and transposed, sorted by C as the accelerator driver sorts its stacks
    order = np.argsort(stack[2::3], kind="stable")
    st = np.empty_like(stack)
    for q in range(3):
        st[q::3] = stack[q::3][order]
    rc, c = run_stack(st, a, b, c0.copy(), m, n, k, L.dbcsr_type_real_8)
    assert rc >= 0
    assert rel_err(c, c_ref) <= 1e-10


@pytest.mark.parametrize("bt", [True, False])
@pytest.mark.parametrize("m,n,k", [(33, 33, 33), (36, 40, 35), (40, 37, 9), (34, 35, 80)])
def test_blocks_of_33_to_40_take_the_one_wave_stack_kernel(m, n, k, bt):
    """round 6: homogeneous stacks of blocks with 33 ... 40 rows and columns (any inner dimension) run smm_stack_f64_mid -- a wave per group of entries, the block
    covered in units of 4 x 4, operand slabs of 8 inner indices --, B transposed or as stored; integer inputs: EXACT"""
    na, nb, nc, nstack = 60, 70, 9, 403
    a = O.mat_init(na, m, k, 42)
    b = O.mat_init(nb, k, n, 24)
    stack = O.stack_init(nstack, nc, na, nb, m, n, k, rseed=13)
    c_ref = np.zeros(nc * m * n)
    O.stack_calc(stack, c_ref, a, b, m, n, k, b_transposed=False)
    rc, c = run_stack(stack, a, b, np.zeros(nc * m * n), m, n, k, L.dbcsr_type_real_8, max_kernel_dim=80 if bt else 0, transpose_b=bt)
    assert rc >= 0
    name = L.load_library().dbcsr_amd_smm_last_kernel().decode()
    assert name.startswith("smm_stack_f64_mid(%d,%d,%d" % (m, n, k)) and ("transposed" in name) == bt, name
    assert np.array_equal(c, c_ref)
