"""TEST-ONLY stand-in for dbcsr_amd.multiply.MultiplyEngine on CPU tensors, built on the oracle.
It lets the distribution / schedule / communication logic of dbcsr_amd.cannon run under the
gloo backend without a GPU.  Never imported by the product."""
import numpy as np
import torch

from dbcsr_amd.matrix import DbcsrMatrix
from oracle import oracle as O


class _Counts:
    def __init__(self, c_nblks=0, c_nze=0, nproducts=0, flop=0):
        self.c_nblks, self.c_nze, self.nproducts, self.flop = c_nblks, c_nze, nproducts, flop


def _to_oracle(M, data=None):
    rs, cs, row_p, col_i, blk_p, d = M.to_host()
    if data is not None:
        d = data
    return O.Bcsr(rs, cs, row_p, col_i, blk_p, np.ascontiguousarray(d, np.float64))


def gathered(M):
    """panel matrices point into a concatenated buffer: re-pack to the oracle's compact layout"""
    rs, cs, rp, ci, bp, d = M.to_host()
    rows = np.repeat(np.arange(len(rs)), np.diff(rp))
    nze = rs[rows].astype(np.int64) * cs[ci].astype(np.int64)
    nbp = np.concatenate([[0], np.cumsum(nze)[:-1]]).astype(np.int64) if len(nze) else np.zeros(0, np.int64)
    nd = np.concatenate([d[bp[b]:bp[b] + nze[b]] for b in range(len(ci))]) if len(ci) else np.zeros(0)
    return O.Bcsr(rs, cs, rp, ci, nbp, nd)


class OracleBackend:
    def fill_random_dist(self, M, counter, row_gid, col_gid, nblkrows_global, stream=None):
        rs, cs, row_p, col_i, blk_p, _ = M.to_host()
        rows = np.repeat(np.arange(len(rs), dtype=np.int32), np.diff(row_p))
        rg, cg = row_gid.numpy(), col_gid.numpy()
        grow, gcol = rg[rows].astype(np.int32), cg[col_i].astype(np.int32)
        rs_g = np.zeros(int(rg.max()) + 1 if len(rg) else 1, np.int32)
        rs_g[rg] = rs
        cs_g = np.zeros(int(cg.max()) + 1 if len(cg) else 1, np.int32)
        cs_g[cg] = cs
        out = np.empty(M.data.numel(), np.float64)
        O.lib().orc_fill_blocks_d(len(grow), np.ascontiguousarray(grow), np.ascontiguousarray(gcol), int(nblkrows_global), 0,
                                  int(counter), rs_g, cs_g, np.ascontiguousarray(blk_p, np.int64), out)
        M.data.copy_(torch.from_numpy(out))

    def symbolic(self, A, B, Cm, retain_sparsity=False, stream=None):
        a = _to_oracle(A, np.zeros(int(A.data_numel) if hasattr(A, "data_numel") else A.data.numel()))
        b = _to_oracle(B, np.zeros(int(B.data_numel) if hasattr(B, "data_numel") else B.data.numel()))
        # pattern-only matrices carry no data: rebuild consistent block offsets for the oracle
        for m in (a, b):
            rows = m.rows()
            nze = m.row_sizes[rows].astype(np.int64) * m.col_sizes[m.col_i].astype(np.int64)
            m.blk_p = np.concatenate([[0], np.cumsum(nze)[:-1]]).astype(np.int64) if len(nze) else np.zeros(0, np.int64)
            m.data = np.zeros(int(nze.sum()))
        out, info = O.multiply("N", "N", 1.0, a, b, 1.0, _to_oracle(Cm), retain_sparsity=retain_sparsity)
        self._last = out
        return torch.from_numpy(out.row_p.copy()), _Counts(out.nblks, len(out.data), info["nproducts"], info["flop"])

    def init_c(self, beta, Cm, row_p, counts, dtype, stream=None):
        s = self._last
        cin = _to_oracle(Cm)
        data = np.zeros(len(s.data))
        pos = {}
        rows = s.rows()
        for b in range(s.nblks):
            pos[(int(rows[b]), int(s.col_i[b]))] = b
        crow = cin.rows()
        for b in range(cin.nblks):
            t = pos[(int(crow[b]), int(cin.col_i[b]))]
            ne = int(cin.row_sizes[crow[b]]) * int(cin.col_sizes[cin.col_i[b]])
            data[s.blk_p[t]:s.blk_p[t] + ne] = beta * cin.data[cin.blk_p[b]:cin.blk_p[b] + ne]
        return DbcsrMatrix(Cm.row_blk_size, Cm.col_blk_size, row_p, torch.from_numpy(s.col_i.copy()), torch.from_numpy(s.blk_p.copy()),
                           torch.from_numpy(data), "C")

    def numeric_after_symbolic(self, alpha, A, B, beta, Cm, row_p, counts, dtype, stream=None):
        out, _ = O.multiply("N", "N", alpha, gathered(A), gathered(B), beta, _to_oracle(Cm))
        return DbcsrMatrix(Cm.row_blk_size, Cm.col_blk_size, torch.from_numpy(out.row_p.copy()), torch.from_numpy(out.col_i.copy()),
                           torch.from_numpy(out.blk_p.copy()), torch.from_numpy(out.data.copy()), "C")

    def multiply_local(self, alpha, A, B, beta, Cm, retain_sparsity=False, stream=None, filter_eps=0.0):
        if retain_sparsity or filter_eps:
            out, info = O.multiply("N", "N", alpha, gathered(A), gathered(B), beta, _to_oracle(Cm), retain_sparsity=retain_sparsity,
                                   filter_eps=filter_eps or 0.0)
            M = DbcsrMatrix(Cm.row_blk_size, Cm.col_blk_size, torch.from_numpy(out.row_p.copy()), torch.from_numpy(out.col_i.copy()),
                            torch.from_numpy(out.blk_p.copy()), torch.from_numpy(out.data.copy()), "C")
            return M, _Counts(out.nblks, len(out.data), info["nproducts"], info["flop"])
        row_p, counts = self.symbolic(A, B, Cm, retain_sparsity)
        info_flop, info_np = counts.flop, counts.nproducts
        out = self.numeric_after_symbolic(alpha, A, B, beta, Cm, row_p, counts, torch.float64)
        return out, _Counts(out.nblks, out.data.numel(), info_np, info_flop)

    def accumulate(self, alpha, A, B, Cacc, stream=None):
        out, info = O.multiply("N", "N", alpha, _to_oracle(A), _to_oracle(B), 1.0, _to_oracle(Cacc), retain_sparsity=True)
        assert np.array_equal(out.col_i, Cacc.col_i.numpy())
        Cacc.data.copy_(torch.from_numpy(out.data))
        return _Counts(out.nblks, len(out.data), info["nproducts"], info["flop"])
