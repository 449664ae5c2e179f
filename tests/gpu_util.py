"""Helpers shared by the -m gpu parity tests (device <-> oracle plumbing)."""

import numpy as np
import torch

from dbcsr_amd import lib as L
from dbcsr_amd.matrix import DbcsrMatrix, StreamHandle


def to_dev(M, name=""):
    return DbcsrMatrix.from_host(M.row_sizes, M.col_sizes, M.row_p, M.col_i, M.blk_p, M.data, name=name)


def dev_to_bcsr(D):
    from oracle import oracle as O
    rs, cs, row_p, col_i, blk_p, data = D.to_host()
    return O.Bcsr(rs, cs, row_p, col_i, blk_p, data)


def rel_err(x, ref):
    x, ref = np.asarray(x, np.float64), np.asarray(ref, np.float64)
    if ref.size == 0:
        return 0.0
    scale = np.maximum(np.abs(ref), 1e-300)
    return float(np.max(np.abs(x - ref) / scale))


def run_stack(stack, a, b, c, m, n, k, dtype_code, max_kernel_dim=80, transpose_b=True):
    """transpose B blocks (as the Fortran host does each tick) then libsmm_acc_process; returns C (host)."""
    lib = L.load_library()
    dev = "cuda"
    st = StreamHandle()
    ta, tb, tc = torch.as_tensor(a).to(dev), torch.as_tensor(b).to(dev), torch.as_tensor(c).to(dev)
    tstack = torch.as_tensor(np.ascontiguousarray(stack, np.int32)).to(dev)
    nstack = len(stack) // 3
    if transpose_b:
        kn = k * n
        trs = torch.arange(0, b.size, kn, dtype=torch.int32, device=dev)
        rc = lib.libsmm_acc_transpose(trs.data_ptr(), 0, int(trs.numel()), tb.data_ptr(), dtype_code, k, n, max_kernel_dim, st.ptr)
        assert rc == 0
    rc = lib.libsmm_acc_process(None, tstack.data_ptr(), nstack, dtype_code, ta.data_ptr(), tb.data_ptr(), tc.data_ptr(), m, n, k,
                                max_kernel_dim, 1, st.ptr, st.ptr)
    torch.cuda.synchronize()
    return rc, tc.cpu().numpy()
