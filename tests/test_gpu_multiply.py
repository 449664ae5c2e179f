"""GPU parity of the device-resident multiply (dbcsr_multiply mirror ->
dbcsr_amd_mm_symbolic/numeric) against the CPU oracle and the reference's golden
checksums.  Bar (north star): C index structure bit-exact, C values within 1e-10
relative of the CPU path."""
import json
import os

import numpy as np
import pytest
import torch

from dbcsr_amd.multiply import MultiplyEngine, dbcsr_multiply
from oracle import oracle as O
from tests.gpu_util import dev_to_bcsr, rel_err, to_dev

pytestmark = pytest.mark.gpu

TOL = 1e-10
CASES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "perf_golden.json")))
CHECKED = sorted(k for k, v in CASES.items() if v["check"] == "T")


def check_against_oracle(A, B, Cm, transa="N", transb="N", alpha=1.0, beta=1.0, retain=False, tol=TOL):
    ref, info = O.multiply(transa, transb, alpha, A, B, beta, Cm, retain_sparsity=retain)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply(transa, transb, alpha, dA, dB, beta, dC, retain_sparsity=retain, flop=flop)
    torch.cuda.synchronize()
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p), "row_p differs"
    assert np.array_equal(out.col_i, ref.col_i), "col_i differs"
    assert np.array_equal(out.blk_p, ref.blk_p), "blk_p differs"
    assert flop[0] == info["flop"]
    assert rel_err(out.data, ref.data) <= tol
    return out, ref


@pytest.mark.parametrize("name", CHECKED)
def test_reference_golden_perf_inputs(name):
    c = CASES[name]
    A, B, Cm = O.perf_case(c["M"], c["N"], c["K"], c["sparsity_a"], c["sparsity_b"], c["sparsity_c"], c["bs_m"], c["bs_n"],
                           c["bs_k"], c["transa"], c["transb"])
    out, _ = check_against_oracle(A, B, Cm, c["transa"], c["transb"], c["alpha"][0], c["beta"][0])
    eng = MultiplyEngine()
    cs, cs_pos = eng.checksum(to_dev(out))
    assert abs(cs / c["checksum"] - 1.0) <= c["threshold"]
    assert abs(cs_pos / c["checksum_pos"] - 1.0) <= c["threshold"]


# GPU-relevant block mixes of the reference's unit tests (tests/dbcsr_unittest3.F:79-120)
MIXES = [[1, 1, 1, 3, 1, 4], [1, 4, 1, 5, 1, 7], [1, 5, 1, 8, 1, 9], [1, 4, 1, 13, 1, 25], [1, 14, 1, 29, 1, 32], [1, 23],
         [1, 45, 1, 67, 1, 78]]


@pytest.mark.parametrize("mix", MIXES)
def test_unittest3_block_mixes(mix):
    A, B, Cm = O.perf_case(300, 260, 280, 0.5, 0.5, 0.5, mix, mix, mix)
    check_against_oracle(A, B, Cm, alpha=1.0, beta=1.0)


@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (-0.5, 2.0), (0.0, 1.0), (3.0, -1.0)])
def test_alpha_beta(alpha, beta):
    A, B, Cm = O.perf_case(230, 184, 207, 0.6, 0.6, 0.4, [1, 23], [1, 23], [1, 23])
    check_against_oracle(A, B, Cm, alpha=alpha, beta=beta)


@pytest.mark.parametrize("ta,tb", [("N", "T"), ("T", "N"), ("T", "T")])
def test_transposes_mixed_sizes(ta, tb):
    A, B, Cm = O.perf_case(150, 170, 190, 0.5, 0.5, 0.7, [1, 13, 2, 5], [1, 23, 1, 4], [2, 7, 1, 32], transa=ta, transb=tb)
    check_against_oracle(A, B, Cm, transa=ta, transb=tb, alpha=0.7, beta=1.3)


def test_retain_sparsity_and_empty_c():
    A, B, Cm = O.perf_case(200, 200, 200, 0.6, 0.6, 0.8, [1, 13, 1, 23], [1, 23, 1, 32], [1, 5, 1, 13])
    check_against_oracle(A, B, Cm, retain=True)
    # C with no blocks at all (beta irrelevant), and retain_sparsity on an empty C -> empty result
    Ce = O.Bcsr(Cm.row_sizes, Cm.col_sizes, np.zeros(Cm.nbr + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int64),
                np.zeros(0))
    check_against_oracle(A, B, Ce, beta=0.0)
    check_against_oracle(A, B, Ce, retain=True)


def test_empty_operands_and_ragged_tail():
    # A empty -> C = beta*C; ragged tails (47 = 2*23 + 1, 31 = 2*13 + 5)
    A, B, Cm = O.perf_case(47, 31, 29, 1.0 - 1e-12, 0.3, 0.5, [1, 23], [1, 13], [1, 7])
    assert A.nblks == 0
    check_against_oracle(A, B, Cm, beta=2.0)
    A, B, Cm = O.perf_case(47, 31, 29, 0.0, 0.0, 1.0 - 1e-12, [1, 23], [1, 13], [1, 7])
    check_against_oracle(A, B, Cm)


def test_h2o_like_config2_shape_small():
    # BASELINE config 2's shape at oracle-friendly size: 23x23 blocks with a 16 tail, 10 % fill
    n = 23 * 150 + 16
    A, B, Cm = O.perf_case(n, n, n, 0.9, 0.9, 0.9, [1, 23], [1, 23], [1, 23])
    check_against_oracle(A, B, Cm)


def test_config3_mixed_13_23_32_with_tail():
    n = 68 * 40 + 24
    A, B, Cm = O.perf_case(n, n, n, 0.95, 0.95, 0.95, [1, 13, 1, 23, 1, 32], [1, 13, 1, 23, 1, 32], [1, 13, 1, 23, 1, 32])
    check_against_oracle(A, B, Cm)


def test_fp32_32x32_config5_shape_small():
    A, B, Cm = O.perf_case(1024, 1024, 1024, 0.8, 0.8, 0.8, [1, 32], [1, 32], [1, 32])
    ref, info = O.multiply("N", "N", 1.0, A, B, 1.0, Cm)
    f32 = lambda M: O.Bcsr(M.row_sizes, M.col_sizes, M.row_p, M.col_i, M.blk_p, M.data.astype(np.float32))
    dA, dB, dC = to_dev(f32(A)), to_dev(f32(B)), to_dev(f32(Cm))
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC)
    torch.cuda.synchronize()
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i)
    assert rel_err(out.data, ref.data) <= 1e-5  # fp32 arithmetic vs fp64 oracle, K ~ 6*32 terms


def test_device_generator_matches_oracle():
    # bench inputs are generated in HBM; the generator must be the reference's (dbcsr_test_methods.F:423-429)
    eng = MultiplyEngine()
    A, B, Cm = O.perf_case(500, 300, 400, 0.7, 0.7, 0.7, [1, 23, 1, 5], [1, 13], [1, 32])
    for M, counter in ((Cm, O.RANDMAT_SEED_INIT + 1), (A, O.RANDMAT_SEED_INIT + 2), (B, O.RANDMAT_SEED_INIT + 3)):
        d = to_dev(M)
        d.data.zero_()
        eng.fill_random(d, counter)
        torch.cuda.synchronize()
        assert np.array_equal(d.data.cpu().numpy(), M.data)
    A32 = O.make_random_matrix(A.row_sizes, A.col_sizes, 0.7, 77, np.float32)
    d = to_dev(A32)
    d.data.zero_()
    eng.fill_random(d, 77)
    torch.cuda.synchronize()
    assert np.array_equal(d.data.cpu().numpy(), A32.data)


def test_checksum_matches_oracle():
    eng = MultiplyEngine()
    A, _, _ = O.perf_case(500, 300, 400, 0.7, 0.7, 0.7, [1, 23, 1, 5], [1, 13], [1, 32])
    cs, csp = eng.checksum(to_dev(A))
    assert abs(cs / O.checksum(A) - 1) < 1e-13 and abs(csp / O.checksum(A, True) - 1) < 1e-13


@pytest.mark.parametrize("nvirt,mode", [(1, "ticks"), (3, "ticks"), (3, "gather")])
def test_cannon_driver_single_gpu_ticks(nvirt, mode):
    # the multi-GPU driver on one rank: structure-once + in-place tick accumulation + distributed generator
    from dbcsr_amd import cannon
    M, N, K, sp = 23 * 30 + 16, 23 * 25 + 16, 23 * 28 + 16, (0.8, 0.8, 0.85)
    grid = cannon.Grid(1, 0, 1, 1, nvirt=nvirt)
    plan = cannon.CannonMultiply(M, N, K, sp, [1, 23], dtype=torch.float64, engine=MultiplyEngine(), grid=grid, mode=mode)
    Cout, counts = plan.multiply(0.5, 2.0)
    torch.cuda.synchronize()
    A, B, Cm = O.perf_case(M, N, K, *sp, [1, 23], [1, 23], [1, 23])
    ref, info = O.multiply("N", "N", 0.5, A, B, 2.0, Cm)
    out = dev_to_bcsr(Cout)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert counts.flop == info["flop"]
    assert rel_err(out.data, ref.data) <= TOL


@pytest.mark.parametrize("symbolic", [None, "rows", "word"])
@pytest.mark.parametrize("eps,retain,alpha", [(2.0, False, 1.0), (40.0, False, 0.5), (60.0, True, 1.0), (1e-30, False, 1.0)])
def test_filter_eps_matches_oracle(eps, retain, alpha, symbolic, monkeypatch):
    # (symbolic: the product-driven / per-word kernel families forced; None = the automatic choice)
    if symbolic:
        monkeypatch.setenv("DBCSR_AMD_MM_SYMBOLIC", symbolic)
    else:
        monkeypatch.delenv("DBCSR_AMD_MM_SYMBOLIC", raising=False)
    # on-the-fly product filter + final block filter (CP2K's linear-scaling mode); eps chosen so that a good part
    # of the products / blocks is actually dropped (U(0,1) 13..23-blocks have squared norms of ~50..180)
    A, B, Cm = O.perf_case(300, 280, 290, 0.5, 0.5, 0.6, [1, 13, 1, 23], [1, 23, 1, 5], [1, 13, 1, 7])
    # spread the block magnitudes so that the norm test separates products
    rng = np.random.default_rng(5)
    for M in (A, B, Cm):
        rows = M.rows()
        for b in range(M.nblks):
            ne = int(M.row_sizes[rows[b]]) * int(M.col_sizes[M.col_i[b]])
            M.data[M.blk_p[b]:M.blk_p[b] + ne] *= 10.0 ** rng.uniform(-2, 0.5)
    ref, info = O.multiply("N", "N", alpha, A, B, 1.0, Cm, retain_sparsity=retain, filter_eps=eps)
    full, info_full = O.multiply("N", "N", alpha, A, B, 1.0, Cm, retain_sparsity=retain)
    if eps > 1:
        assert info["nproducts"] < info_full["nproducts"]  # the filter really removed products
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", alpha, dA, dB, 1.0, dC, retain_sparsity=retain, filter_eps=eps, flop=flop, engine=MultiplyEngine())
    torch.cuda.synchronize()
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert flop[0] == info["flop"]
    assert rel_err(out.data, ref.data) <= TOL


@pytest.mark.parametrize("mix", [[1, 13, 1, 23, 1, 32], [1, 5, 1, 7, 1, 9], [1, 31], [1, 1, 1, 3, 1, 4]])
def test_fp32_mixed_sizes(mix):
    # the fp32 kernel stages B transposed in LDS with a multiply-shift division by the runtime k: odd sizes matter
    A, B, Cm = O.perf_case(330, 310, 290, 0.6, 0.6, 0.6, mix, mix, mix)
    ref, info = O.multiply("N", "N", 0.5, A, B, 2.0, Cm)
    f32 = lambda M: O.Bcsr(M.row_sizes, M.col_sizes, M.row_p, M.col_i, M.blk_p, M.data.astype(np.float32))
    dA, dB, dC = to_dev(f32(A)), to_dev(f32(B)), to_dev(f32(Cm))
    flop = [0]
    dbcsr_multiply("N", "N", 0.5, dA, dB, 2.0, dC, flop=flop)
    torch.cuda.synchronize()
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert flop[0] == info["flop"]
    assert float(np.max(np.abs(out.data - ref.data))) <= 1e-5 * float(np.max(np.abs(ref.data)))


@pytest.mark.parametrize("bs_m,bs_n,bs_k", [([1, 4], [1, 4], [1, 4]), ([1, 1, 1, 2, 1, 3, 1, 4], [1, 4, 1, 3, 1, 1], [1, 4]),
                                            ([1, 3], [1, 2], [1, 9, 1, 5, 1, 13]), ([1, 4], [1, 4], [1, 32, 1, 1])])
@pytest.mark.parametrize("alpha,beta,retain", [(1.0, 1.0, False), (-0.5, 2.0, False), (1.0, 0.0, True)])
def test_blocks_of_at_most_4x4_packed_kernel(bs_m, bs_n, bs_k, alpha, beta, retain):
    """C blocks of at most 4 x 4 run the packed kernel (four C blocks per wavefront); k is unrestricted."""
    A, B, Cm = O.perf_case(150, 130, 170, 0.7, 0.6, 0.8, bs_m, bs_n, bs_k)
    check_against_oracle(A, B, Cm, alpha=alpha, beta=beta, retain=retain)


def test_config1_shape_4x4_blocks():
    A, B, Cm = O.perf_case(512, 512, 512, 0.9, 0.9, 0.9, [1, 4], [1, 4], [1, 4])
    check_against_oracle(A, B, Cm)


@pytest.mark.parametrize("nchunks", [2, 3, 7])
def test_k_chunked_passes_match_oracle(nchunks):
    """L2 blocking over k (MultiplyEngine.multiply_local(kchunks=n)): same structure, values within tolerance."""
    A, B, Cm = O.perf_case(260, 240, 300, 0.5, 0.6, 0.7, [1, 13, 1, 5], [1, 23, 1, 4], [1, 7, 1, 32, 1, 9])
    ref, info = O.multiply("N", "N", -1.5, A, B, 0.5, Cm)
    E = MultiplyEngine()
    out, counts = E.multiply_local(-1.5, to_dev(A), to_dev(B), 0.5, to_dev(Cm), kchunks=nchunks)
    torch.cuda.synchronize()
    got = dev_to_bcsr(out)
    assert np.array_equal(got.row_p, ref.row_p) and np.array_equal(got.col_i, ref.col_i) and np.array_equal(got.blk_p, ref.blk_p)
    assert counts.flop == info["flop"] and counts.nproducts == info["nproducts"]
    assert np.all(np.abs(got.data - ref.data) <= 1e-10 * np.maximum(np.abs(ref.data), 1.0))


def test_k_chunked_passes_reuse_their_views_and_plans():
    """the passes' operands are index views over the operands' data areas, one engine per pass keeps its plan: repeated multiplies of
    the same matrices (and of new VALUES in the same arrays) reuse both and stay correct; another pattern rebuilds them"""
    A, B, Cm = O.perf_case(23 * 30 + 16, 23 * 28 + 16, 23 * 34 + 16, 0.6, 0.6, 0.7, [1, 23], [1, 23], [1, 23])
    ref, info = O.multiply("N", "N", 0.5, A, B, 1.5, Cm)
    E = MultiplyEngine()
    E.trust_plan(True)
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    for rep in range(3):
        out, counts = E.multiply_local(0.5, dA, dB, 1.5, dC, kchunks=4)
        torch.cuda.synchronize()
        got = dev_to_bcsr(out)
        assert np.array_equal(got.col_i, ref.col_i) and np.array_equal(got.blk_p, ref.blk_p) and counts.flop == info["flop"]
        assert np.all(np.abs(got.data - ref.data) <= 1e-10 * np.maximum(np.abs(ref.data), 1.0))
    assert len(set(id(e) for _, _, e in E._kpass)) == len(E._kpass), "plenty of memory here: one engine per pass"
    built = [e.plan_stats() for _, _, e in E._kpass]
    assert all(b == (2, 1) for b in built), built   # every pass: one plan built, reused twice
    dA.data.mul_(-2.0)   # new values in the same arrays: same views, same plans
    ref2, _ = O.multiply("N", "N", 0.5, O.Bcsr(A.row_sizes, A.col_sizes, A.row_p, A.col_i, A.blk_p, A.data * -2.0), B, 1.5, Cm)
    out, _ = E.multiply_local(0.5, dA, dB, 1.5, dC, kchunks=4)
    torch.cuda.synchronize()
    assert np.all(np.abs(dev_to_bcsr(out).data - ref2.data) <= 1e-10 * np.maximum(np.abs(ref2.data), 1.0))
    assert all(e.plan_stats() == (3, 1) for _, _, e in E._kpass)
    A3, B3, C3 = O.perf_case(23 * 30 + 16, 23 * 28 + 16, 23 * 34 + 16, 0.5, 0.7, 0.6, [1, 23], [1, 23], [1, 23])
    ref3, _ = O.multiply("N", "N", 0.5, A3, B3, 1.5, C3)
    out, _ = E.multiply_local(0.5, to_dev(A3), to_dev(B3), 1.5, to_dev(C3), kchunks=4)
    torch.cuda.synchronize()
    got = dev_to_bcsr(out)
    assert np.array_equal(got.col_i, ref3.col_i) and np.all(np.abs(got.data - ref3.data) <= 1e-10 * np.maximum(np.abs(ref3.data), 1.0))


@pytest.mark.parametrize("size", list(range(9, 33)))
def test_exact_size_kernels_every_cube_with_tails(size):
    """Uniform size s (one exact-size kernel per cube 9..32) with a ragged tail block in every dimension: the tail row / column /
    inner blocks take the fall-back paths inside the same launch.  fp64 at 1e-10, fp32 against the fp64 oracle at 1e-5."""
    dim = 12 * size + max(1, size // 3)
    A, B, Cm = O.perf_case(dim, dim, dim, 0.6, 0.6, 0.7, [1, size], [1, size], [1, size])
    check_against_oracle(A, B, Cm, alpha=0.75, beta=-1.25)
    ref, info = O.multiply("N", "N", 0.75, A, B, -1.25, Cm)
    f32 = lambda M: O.Bcsr(M.row_sizes, M.col_sizes, M.row_p, M.col_i, M.blk_p, M.data.astype(np.float32))
    dA, dB, dC = to_dev(f32(A)), to_dev(f32(B)), to_dev(f32(Cm))
    dbcsr_multiply("N", "N", 0.75, dA, dB, -1.25, dC)
    torch.cuda.synchronize()
    out = dev_to_bcsr(dC)
    assert np.array_equal(out.row_p, ref.row_p) and np.array_equal(out.col_i, ref.col_i) and np.array_equal(out.blk_p, ref.blk_p)
    assert float(np.max(np.abs(out.data - ref.data))) <= 1e-5 * float(np.max(np.abs(ref.data)))


def test_two_engines_on_two_streams_concurrently():
    """One engine handle per stream (the reference gives every OpenMP thread its own stream): two multiplies in flight at once."""
    import threading
    cases = [O.perf_case(460, 460, 460, 0.6, 0.6, 0.6, [1, 23], [1, 23], [1, 23]),
             O.perf_case(390, 420, 400, 0.5, 0.5, 0.5, [1, 13, 1, 5], [1, 32, 1, 7], [1, 9, 1, 23])]
    refs = [O.multiply("N", "N", 1.0, A, B, 1.0, Cm)[0] for A, B, Cm in cases]
    outs, errs = [None, None], []

    def work(i):
        try:
            st = torch.cuda.Stream()
            E = MultiplyEngine()
            A, B, Cm = cases[i]
            with torch.cuda.stream(st):
                dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
                for _ in range(5):
                    out, _ = E.multiply_local(1.0, dA, dB, 1.0, dC, stream=st)
                st.synchronize()
                outs[i] = dev_to_bcsr(out)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for got, ref in zip(outs, refs):
        assert np.array_equal(got.row_p, ref.row_p) and np.array_equal(got.col_i, ref.col_i)
        assert rel_err(got.data, ref.data) <= TOL


def test_mnk_statistics_match_enumeration():
    # per-(m, n, k) product and flop counts (reference: dbcsr_mm_sched.F:392-461) against a direct enumeration of the products
    A, B, Cm = O.perf_case(300, 280, 260, 0.5, 0.5, 0.6, [1, 13, 1, 23, 1, 32, 1, 7], [1, 23, 1, 5, 1, 32], [1, 13, 1, 32, 1, 9])
    eng = MultiplyEngine()
    dA, dB, dC = to_dev(A), to_dev(B), to_dev(Cm)
    flop = [0]
    dbcsr_multiply("N", "N", 1.0, dA, dB, 1.0, dC, flop=flop, engine=eng)
    got = {(m, n, k): (c, f) for m, n, k, c, f in eng.mnk_statistics()}
    want = {}
    rows = A.rows()
    for ab in range(A.nblks):
        i, kb = rows[ab], A.col_i[ab]
        for bb in range(B.row_p[kb], B.row_p[kb + 1]):
            key = (int(A.row_sizes[i]), int(B.col_sizes[B.col_i[bb]]), int(A.col_sizes[kb]))
            c, f = want.get(key, (0, 0))
            want[key] = (c + 1, f + 2 * key[0] * key[1] * key[2])
    assert got == want
    assert sum(f for _, f in got.values()) == flop[0]
    tot = eng.accumulate_statistics()
    assert tot == got
    import io
    buf = io.StringIO()
    eng.print_statistics(buf)
    assert "flops total" in buf.getvalue() and "100.0%" in buf.getvalue()


@pytest.mark.parametrize("symm", ["S", "A"])
def test_symmetric_product_is_the_triangle_of_the_full_product(symm):
    """A product matrix with symmetry at a size where the exact-size kernel runs (8192^2, 23 x 23 blocks, 10 %): C_sym = A A^T (symmetric)
    or A B^T - B A^T (antisymmetric) computed on the stored triangle only must equal the triangle of the full product, with about
    half its products."""
    from dbcsr_amd.matrix import DbcsrMatrix
    from dbcsr_amd.randmat import perf_matrices
    E = MultiplyEngine()
    A, B, _ = perf_matrices(8192, 8192, 8192, (0.9, 0.9, 0.9), [1, 23], [1, 23], [1, 23], dtype=torch.float64, engine=E)
    dev = A.data.device
    empty = lambda: DbcsrMatrix.empty_like_pattern(A.row_blk_size, A.row_blk_size, torch.float64, device=dev)
    full = empty()
    if symm == "S":
        c_full = dbcsr_multiply("N", "T", 1.0, A, A, 0.0, full, engine=E)
    else:
        dbcsr_multiply("N", "T", 1.0, A, B, 0.0, full, engine=E)
        c_full = dbcsr_multiply("N", "T", -1.0, B, A, 1.0, full, engine=E)
    sym = empty()
    sym.symmetry = symm
    if symm == "S":
        c_sym = dbcsr_multiply("N", "T", 1.0, A, A, 0.0, sym, engine=E)
    else:
        dbcsr_multiply("N", "T", 1.0, A, B, 0.0, sym, engine=E)
        c_sym = dbcsr_multiply("N", "T", -1.0, B, A, 1.0, sym, engine=E)
    torch.cuda.synchronize()
    assert "hot<23,23,23>" in E.last_kernel()
    F, S = dev_to_bcsr(full), dev_to_bcsr(sym)
    frow = F.rows()
    keep = np.nonzero(frow <= F.col_i)[0]
    assert np.array_equal(S.rows(), frow[keep]) and np.array_equal(S.col_i, F.col_i[keep])
    nze = F.row_sizes[frow[keep]].astype(np.int64) * F.col_sizes[F.col_i[keep]]
    want = np.concatenate([F.data[F.blk_p[b]:F.blk_p[b] + n] for b, n in zip(keep, nze)])
    assert S.data.size == want.size and rel_err(S.data, want) <= 1e-10
    nb = len(F.row_sizes)
    assert abs(c_sym.flop / c_full.flop - 0.5) < 1.5 / nb + 0.01   # half the products (the diagonal blocks are computed once in both)


def test_filtered_sparse_product_takes_the_product_driven_kernels():
    """CP2K's linear-scaling regime: sparse operands AND filter_eps.  At 4096 block rows and 1 % fill the automatic choice must be the
    product-driven symbolic kernels (also for the filtered C pattern), with the same result as the candidate-driven ones."""
    from dbcsr_amd.randmat import perf_matrices
    res = {}
    for symbolic in ("grid", None):
        if symbolic:
            os.environ["DBCSR_AMD_MM_SYMBOLIC"] = symbolic
        else:
            os.environ.pop("DBCSR_AMD_MM_SYMBOLIC", None)
        try:
            E = MultiplyEngine()
            A, B, Cm = perf_matrices(4096 * 5, 4096 * 5, 4096 * 5, (0.99, 0.99, 0.99), [1, 5], [1, 5], [1, 5], dtype=torch.float64, engine=E)
            flop = [0]
            dbcsr_multiply("N", "N", 1.0, A, B, 1.0, Cm, filter_eps=3.0, flop=flop, engine=E)
            torch.cuda.synchronize()
            res[symbolic] = (dev_to_bcsr(Cm), flop[0])
        finally:
            os.environ.pop("DBCSR_AMD_MM_SYMBOLIC", None)
    (g, fg), (r, fr) = res["grid"], res[None]
    assert fg == fr and np.array_equal(g.row_p, r.row_p) and np.array_equal(g.col_i, r.col_i)
    assert rel_err(r.data, g.data) <= 1e-12
    assert 0 < r.nblks


def test_h2o_like_80_percent_fill_takes_k_passes_by_itself(monkeypatch):
    """The reference's tests/inputs/test_H2O.perf shape (2208^2, 23 x 23 blocks, sparsity 0.2 = 80 % fill).  What ships for dense fills is
    not an operand-sharing kernel but passes over k (profiles/r05_fill_sweep.txt: 0.48-0.50 of the peak from 20 % to 80 % fill, the sharing
    dataflows of the lab build within 1 %); the choice is automatic, by the size of A's block rows.  At this size a block row of A has 325 KB,
    so the 1 MB rule is scaled down for the test: the shipping build must then split the product by itself, and match the oracle either way."""
    from dbcsr_amd import multiply as MM
    for k in ("DBCSR_AMD_MM_KCHUNKS", "DBCSR_AMD_MM_KERNEL", "DBCSR_AMD_MM_HOT"):
        monkeypatch.delenv(k, raising=False)
    A, B, Cm = O.perf_case(2208, 2208, 2208, 0.2, 0.2, 0.2, [1, 23], [1, 23], [1, 23])
    ref, info = O.multiply("N", "N", 1.0, A, B, 1.0, Cm)
    for row_bytes, passes in ((MM.MultiplyEngine.KCHUNK_ROW_BYTES, 1), (100 * 1024.0, 4)):
        monkeypatch.setattr(MM.MultiplyEngine, "KCHUNK_ROW_BYTES", row_bytes)
        E = MultiplyEngine(lab=False)
        out, counts = E.multiply_local(1.0, to_dev(A), to_dev(B), 1.0, to_dev(Cm))
        torch.cuda.synchronize()
        assert E.last_kchunks == passes and E.last_kernel() == "mm_numeric_f64_hot<23,23,23>", (E.last_kchunks, E.last_kernel())
        got = dev_to_bcsr(out)
        assert np.array_equal(got.row_p, ref.row_p) and np.array_equal(got.col_i, ref.col_i)
        assert counts.flop == info["flop"] and counts.nproducts == info["nproducts"]
        assert np.all(np.abs(got.data - ref.data) <= 1e-10 * np.maximum(np.abs(ref.data), 1.0))
