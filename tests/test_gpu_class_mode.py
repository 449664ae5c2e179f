"""When the (m, n)-class kernels (run-time compiled exact-size kernels, one launch per class: the reference compiles one kernel per (m, n, k) too,
src/acc/libsmm_acc/libsmm_acc.cpp:90-195) are chosen WITHOUT being forced, on multiplies with at least 200 000 C blocks -- round 6:
  * uniform RECTANGULAR triplets (a dominant (m, n, k) that is not a cube has no ahead-of-time kernel: one class with one inner size),
  * two block sizes ALTERNATING (a size pattern whose period divides 8: the rows of a class are dealt to the XCDs class by class, mm_symbolic.h: class_row_deal),
  * a cube of 9 ... 32 keeps its ahead-of-time kernel, a cube below 9 the one-tile kernel.
Against the CPU oracle: index bit-exact, flop equal, values 1e-10 relative."""
import numpy as np
import pytest
import torch

from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply
from oracle import oracle as O
from tests.gpu_util import dev_to_bcsr, rel_err, to_dev

pytestmark = pytest.mark.gpu

ENV = ("DBCSR_AMD_MM_KERNEL", "DBCSR_AMD_MM_HOT", "DBCSR_AMD_MM_TINY", "DBCSR_AMD_MM_SYMBOLIC", "DBCSR_AMD_MM_CLASSES", "DBCSR_AMD_MM_WG_WAVES",
       "DBCSR_AMD_MM_SMALL", "DBCSR_AMD_MM_KCHUNKS", "DBCSR_AMD_MM_WORK", "DBCSR_AMD_MM_MID", "DBCSR_AMD_MM_BIG")

# (M, N, K, sparsity A, B, C, mix m, mix n, mix k), expected kernel (prefix)
CASES = {
    "2x9x3": ((2 * 470, 9 * 450, 3 * 100, 0.7, 0.7, 0.5, [1, 2], [1, 9], [1, 3]), "mm_numeric_f64_class[1 jit + 0 generic"),
    "13x5x23": ((13 * 470, 5 * 450, 23 * 40, 0.6, 0.6, 0.5, [1, 13], [1, 5], [1, 23]), "mm_numeric_f64_class[1 jit + 0 generic"),
    "9x32x9_tails": ((9 * 470 + 4, 32 * 450 + 7, 9 * 60 + 5, 0.6, 0.6, 0.5, [1, 9], [1, 32], [1, 9]), "mm_numeric_f64_class["),
    "alternating_5_13": ((18 * 240, 18 * 235, 18 * 25, 0.6, 0.6, 0.5, [1, 5, 1, 13], [1, 5, 1, 13], [1, 5, 1, 13]), "mm_numeric_f64_class[4 jit + 0 generic"),
    "period_4": ((28 * 120, 28 * 118, 28 * 14, 0.6, 0.6, 0.5, [1, 5, 1, 9, 1, 5, 1, 9], [1, 9, 1, 5], [1, 5, 2, 9, 1, 5]), "mm_numeric_f64_class[4 jit + 0 generic"),
    "alternating_3_13_rows_only": ((16 * 250, 7 * 480, 11 * 40, 0.6, 0.6, 0.5, [1, 3, 1, 13], [1, 7], [1, 11]), "mm_numeric_f64_class[2 jit + 0 generic"),
    "cube_13": ((13 * 470, 13 * 450, 13 * 60, 0.6, 0.6, 0.5, [1, 13], [1, 13], [1, 13]), "mm_numeric_f64_hot<13,13,13>"),
    "cube_6": ((6 * 470, 6 * 450, 6 * 100, 0.7, 0.7, 0.5, [1, 6], [1, 6], [1, 6]), "mm_numeric_f64_small<2>"),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_class_mode_is_chosen_and_matches_oracle(monkeypatch, name):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    case, expect = CASES[name]
    A, B, Cm = O.perf_case(*case)
    alpha, beta = 0.6, 1.4
    ref, info = O.multiply("N", "N", alpha, A, B, beta, Cm)
    assert ref.nblks >= 200000, ref.nblks   # (the case must reach the threshold the engine applies)
    eng = MultiplyEngine()
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", alpha, dA, dB, beta, dC, flop=flop, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith(expect), (eng.last_kernel(), expect)
    assert flop[0] == info["flop"]
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert rel_err(out.data, ref.data) <= 1e-10
    # the same plan again, in place (skip_empty) -- the launch order is reused
    ref2, _ = O.multiply("N", "N", -0.5, A, B, 1.0, ref, retain_sparsity=True)
    dbcsr_multiply("N", "N", -0.5, dA, dB, 1.0, dC, retain_sparsity=True, engine=eng)
    torch.cuda.synchronize()
    out2 = dev_to_bcsr(dC)
    assert np.array_equal(out2.col_i, ref2.col_i) and rel_err(out2.data, ref2.data) <= 1e-10


def test_jit_cache_directory_is_filled_and_used(tmp_path):
    """DBCSR_AMD_JIT_CACHE=<dir>: the first process compiles the class kernels and leaves their code objects there, the second loads them (no hiprtc
    compile: its log says so), a truncated file is compiled again and written over; the results of all three agree with the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np, torch\n"
        "from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply\n"
        "from oracle import oracle as O\n"
        "from tests.gpu_util import dev_to_bcsr, rel_err, to_dev\n"
        "A, B, Cm = O.perf_case(18 * 30, 18 * 28, 18 * 12, 0.5, 0.5, 0.5, [1, 5, 1, 13], [1, 13, 1, 5], [1, 5, 1, 13])\n"
        "ref, info = O.multiply('N', 'N', 1.0, A, B, 1.0, Cm)\n"
        "eng = MultiplyEngine(); dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)\n"
        "dbcsr_multiply('N', 'N', 1.0, dA, dB, 1.0, dC, engine=eng); torch.cuda.synchronize()\n"
        "out = dev_to_bcsr(dC)\n"
        "assert eng.last_kernel().startswith('mm_numeric_f64_class[4 jit'), eng.last_kernel()\n"
        "assert np.array_equal(out.col_i, ref.col_i) and rel_err(out.data, ref.data) <= 1e-10\n"
        "print('multiply ok')\n")
    env = dict(os.environ, DBCSR_AMD_JIT_CACHE=str(tmp_path), DBCSR_AMD_MM_CLASSES="2", DBCSR_AMD_MM_VERBOSE="1", PYTHONPATH=root)

    def run():
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "multiply ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
        return r.stderr

    log1 = run()
    files = sorted(p for p in os.listdir(tmp_path) if p.endswith(".co"))
    assert len(files) == 4 and log1.count("compiled class kernel") == 4 and " from " + str(tmp_path) not in log1, (files, log1[-1500:])
    log2 = run()
    assert log2.count("compiled class kernel") == 0 and log2.count(" from " + str(tmp_path)) == 4, log2[-1500:]
    victim = os.path.join(tmp_path, files[0])
    with open(victim, "r+b") as f:
        f.truncate(100)
    log3 = run()
    assert log3.count("compiled class kernel") == 1 and log3.count(" from " + str(tmp_path)) == 3, log3[-1500:]
    assert os.path.getsize(victim) > 1000 and not [p for p in os.listdir(tmp_path) if p.endswith(".tmp")]


# A filtered multiply ends with the block filter on C's norms: the product kernels leave them behind (round 6: also the blocks of another size inside the exact-size
# kernel's launch, the slab kernels and the slab classes of a mixed-size multiply) -- every case against the oracle's filtered product, the chosen kernel asserted
FILTER_CASES = {
    "hot23_tails": ((23 * 20 + 16, 23 * 18 + 9, 23 * 22 + 5, 0.6, 0.6, 0.6, [1, 23], [1, 23], [1, 23]), {}, "mm_numeric_f64_hot<23,23,23>"),
    "classes_13_23_32": ((68 * 6, 68 * 5 + 13, 68 * 6 + 23, 0.6, 0.6, 0.6, [1, 13, 1, 23, 1, 32], [1, 32, 1, 13, 1, 23], [1, 23, 1, 32, 1, 13]),
                         {"DBCSR_AMD_MM_CLASSES": "2"}, "mm_numeric_f64_class[6 jit + 3 slab"),
    "mid36_tail": ((36 * 9 + 20, 36 * 8 + 7, 36 * 9 + 30, 0.6, 0.6, 0.6, [1, 36], [1, 36], [1, 36]), {}, "mm_numeric_f64_mid<9,9>"),
    "mid_33_36": ((69 * 5, 69 * 5 + 33, 69 * 4, 0.6, 0.6, 0.6, [1, 33, 1, 36], [1, 36, 1, 33], [1, 33, 1, 36]), {}, "mm_numeric_f64_mid<9,9>"),
    "mid_30_40": ((70 * 5, 70 * 5 + 30, 70 * 4, 0.6, 0.6, 0.6, [1, 30, 1, 40], [1, 40, 1, 30], [1, 30, 1, 40]), {}, "mm_numeric_f64_mid<10,10>"),
}


@pytest.mark.parametrize("name", sorted(FILTER_CASES))
@pytest.mark.parametrize("eps", [150.0, 400.0, 1000.0])
def test_filtered_multiply_with_the_kernels_norms(monkeypatch, name, eps):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    case, env, expect = FILTER_CASES[name]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    A, B, Cm = O.perf_case(*case)
    ref, info = O.multiply("N", "N", 1.0, A, B, 0.5, Cm, filter_eps=eps)
    full, _ = O.multiply("N", "N", 1.0, A, B, 0.5, Cm)
    assert ref.nblks < full.nblks, "the filter case does not filter"
    eng = MultiplyEngine()
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", 1.0, dA, dB, 0.5, dC, filter_eps=eps, flop=flop, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith(expect), (eng.last_kernel(), expect)
    assert flop[0] == info["flop"]
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert rel_err(out.data, ref.data) <= 1e-10


# A walk over (m, n) classes at full scale (>= 200 000 C blocks each: an event that hits one block in 500 shows; the small cases of the random sweep would miss it --
# round 6: a variant of the class kernels' epilogue returned wrong LOW bits in 0.2 % of the (9, 32) blocks and only this scale caught it).  Off by default beyond the
# first four shapes: DBCSR_AMD_CLASS_SHAPES=N takes the first N of the list.
SHAPES = [(9, 32, 9), (32, 9, 13), (13, 23, 5), (23, 13, 32), (5, 32, 23), (32, 5, 7), (16, 16, 9), (10, 30, 17), (30, 10, 6), (7, 29, 12), (17, 17, 17), (12, 24, 8),
          (24, 12, 31), (11, 27, 10), (8, 32, 16), (32, 8, 24), (6, 31, 14), (15, 20, 11), (20, 15, 26), (18, 14, 19), (9, 9, 32), (14, 22, 5), (22, 14, 29), (31, 10, 9)]


@pytest.mark.parametrize("shape", SHAPES[:int(__import__("os").environ.get("DBCSR_AMD_CLASS_SHAPES", "4"))], ids=lambda s: "%dx%dx%d" % s)
def test_class_kernels_at_scale_with_tails(monkeypatch, shape):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    m, n, k = shape
    tail = lambda s, salt: 1 + (s * 7 + salt) % (s - 1)   # a tail block of another size in every dimension: more classes, a second inner size
    A, B, Cm = O.perf_case(m * 470 + tail(m, 1), n * 450 + tail(n, 2), k * 60 + tail(k, 3), 0.6, 0.6, 0.5, [1, m], [1, n], [1, k])
    alpha, beta = 0.6, 1.4
    ref, info = O.multiply("N", "N", alpha, A, B, beta, Cm)
    assert ref.nblks >= 200000, ref.nblks
    eng = MultiplyEngine()
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", alpha, dA, dB, beta, dC, flop=flop, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith("mm_numeric_f64_hot<" if m == n == k else "mm_numeric_f64_class["), eng.last_kernel()
    assert flop[0] == info["flop"]
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert rel_err(out.data, ref.data) <= 1e-12


# The same at scale for the other kernel families (exact-size cubes, slab kernels, one-tile kernel): a tail block in every dimension, tens to hundreds of thousands of
# C blocks, 1e-12 against the oracle.  (nbr, nbc, nbk) chosen so that the CPU oracle takes seconds.
FAMILY_SHAPES = [((13, 13, 13), (470, 450, 60), "mm_numeric_f64_hot<13,13,13>"), ((23, 23, 23), (330, 320, 50), "mm_numeric_f64_hot<23,23,23>"),
                 ((32, 32, 32), (250, 240, 40), "mm_numeric_f64_hot<32,32,32>"), ((36, 36, 36), (200, 190, 30), "mm_numeric_f64_mid<9,9>"),
                 ((40, 33, 37), (180, 190, 30), "mm_numeric_f64_mid<10,9>"), ((5, 5, 5), (470, 450, 100), "mm_numeric_f64_small<2>"),
                 ((8, 7, 6), (470, 450, 100), "mm_numeric_f64_small<2>"), ((4, 4, 4), (470, 450, 100), "mm_numeric_f64_tiny")]


@pytest.mark.parametrize("shape,counts,expect", FAMILY_SHAPES, ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) and len(v) == 3 and isinstance(v[0], int) and v[0] < 100 else None)
def test_kernel_families_at_scale_with_tails(monkeypatch, shape, counts, expect):
    for k in ENV:
        monkeypatch.delenv(k, raising=False)
    m, n, k = shape
    nbr, nbc, nbk = counts
    tail = lambda s, salt: 1 + (s * 7 + salt) % (s - 1)
    A, B, Cm = O.perf_case(m * nbr + tail(m, 1), n * nbc + tail(n, 2), k * nbk + tail(k, 3), 0.6, 0.6, 0.5, [1, m], [1, n], [1, k])
    alpha, beta = 0.6, 1.4
    ref, info = O.multiply("N", "N", alpha, A, B, beta, Cm)
    eng = MultiplyEngine()
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", alpha, dA, dB, beta, dC, flop=flop, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith(expect), (eng.last_kernel(), expect)
    assert flop[0] == info["flop"]
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert rel_err(out.data, ref.data) <= 1e-12


def test_announced_filter_leaves_dropped_blocks_unwritten_and_refuses_a_smaller_eps(monkeypatch):
    """dbcsr_amd_mm_expect_filter (include/dbcsr_amd_mm.h): the filtered product equals the one computed without the announcement and the oracle's; the C ABI refuses to
    filter the unannounced-for way (a smaller eps) afterwards, because blocks below the announced threshold were never written."""
    import ctypes as C
    from dbcsr_amd import lib as _lib
    for k in ENV + ("DBCSR_AMD_MM_EXPECT_FILTER",):
        monkeypatch.delenv(k, raising=False)
    case = (23 * 40 + 16, 23 * 38 + 9, 23 * 30 + 5, 0.8, 0.8, 0.97, [1, 23], [1, 23], [1, 23])   # sparse C_in: most C blocks are new
    A, B, Cm = O.perf_case(*case)
    eps = 200.0   # keeps 489 of 1093 blocks
    ref, info = O.multiply("N", "N", 1.0, A, B, 1.0, Cm, filter_eps=eps)
    full, _ = O.multiply("N", "N", 1.0, A, B, 1.0, Cm)
    assert 0 < ref.nblks < 0.8 * full.nblks, (ref.nblks, full.nblks)
    outs = []
    for sw in (None, "0"):
        if sw:
            monkeypatch.setenv("DBCSR_AMD_MM_EXPECT_FILTER", sw)
        eng = MultiplyEngine()
        dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
        dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, filter_eps=eps, engine=eng)
        torch.cuda.synchronize()
        assert eng.last_kernel() == "mm_numeric_f64_hot<23,23,23>"
        out = dev_to_bcsr(dC)
        assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
        assert rel_err(out.data, ref.data) <= 1e-10
        outs.append(out)
    assert np.array_equal(outs[0].data, outs[1].data)
    # the C ABI: announce, multiply, then ask for a SMALLER filter than announced
    monkeypatch.delenv("DBCSR_AMD_MM_EXPECT_FILTER", raising=False)
    eng = MultiplyEngine()
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    L = eng.L
    from dbcsr_amd.matrix import StreamHandle
    sth = StreamHandle(None)
    st = sth.ptr
    a, b, cin = dA.desc(), dB.desc(), dC.desc()
    row_p = torch.empty(dC.nblkrows + 1, dtype=torch.int32, device="cuda")
    counts = _lib.MmCounts()
    assert L.dbcsr_amd_mm_symbolic_filtered(eng.h, dA.dtype_code, 1.0, eps, C.byref(a), C.byref(b), C.byref(cin), 0, row_p.data_ptr(), C.byref(counts), st) == 0
    from dbcsr_amd.matrix import DbcsrMatrix
    out = DbcsrMatrix(dC.row_blk_size, dC.col_blk_size, row_p, torch.empty(counts.c_nblks, dtype=torch.int32, device="cuda"),
                      torch.empty(counts.c_nblks, dtype=torch.int64, device="cuda"), torch.empty(counts.c_nze, dtype=torch.float64, device="cuda"), "C")
    cout = out.desc(out=True)
    assert L.dbcsr_amd_mm_expect_filter(eng.h, eps) == 0
    assert L.dbcsr_amd_mm_numeric(eng.h, dA.dtype_code, 1.0, C.byref(a), C.byref(b), 1.0, C.byref(cin), C.byref(cout), st) == 0
    new_row_p = torch.empty(dC.nblkrows + 1, dtype=torch.int32, device="cuda")
    nb, nz = C.c_int64(0), C.c_int64(0)
    src = out.desc()
    assert L.dbcsr_amd_bcsr_filter_count(eng.h, out.dtype_code, C.byref(src), 0.5 * eps, new_row_p.data_ptr(), C.byref(nb), C.byref(nz), st) == -3


# The announced final filter at scale: tens to hundreds of thousands of C blocks, a third to two thirds of them dropped (and therefore never written), index bit-exact
# and values 1e-10 against the oracle's filtered product
@pytest.mark.parametrize("shape,counts,sp,eps,expect", [
    ((23, 23, 23), (330, 320, 50), (0.8, 0.8, 0.97), 200.0, "mm_numeric_f64_hot<23,23,23>"),
    ((9, 32, 13), (470, 450, 60), (0.75, 0.75, 0.97), 200.0, "mm_numeric_f64_class["),
    ((36, 36, 36), (200, 190, 30), (0.7, 0.7, 0.97), 800.0, "mm_numeric_f64_mid<9,9>"),
], ids=["hot23", "class9x32", "mid36"])
def test_announced_filter_at_scale(monkeypatch, shape, counts, sp, eps, expect):
    for k in ENV + ("DBCSR_AMD_MM_EXPECT_FILTER",):
        monkeypatch.delenv(k, raising=False)
    m, n, k = shape
    nbr, nbc, nbk = counts
    tail = lambda s, salt: 1 + (s * 7 + salt) % (s - 1)
    A, B, Cm = O.perf_case(m * nbr + tail(m, 1), n * nbc + tail(n, 2), k * nbk + tail(k, 3), *sp, [1, m], [1, n], [1, k])
    ref, info = O.multiply("N", "N", 1.0, A, B, 1.0, Cm, filter_eps=eps)
    eng = MultiplyEngine()
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, filter_eps=eps, flop=flop, engine=eng)
    torch.cuda.synchronize()
    assert eng.last_kernel().startswith(expect), (eng.last_kernel(), expect)
    assert flop[0] == info["flop"]
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert rel_err(out.data, ref.data) <= 1e-10
