"""Device-resident BCSR matrix: the host-side mirror of the reference's
``dbcsr_type`` index (src/core/dbcsr_types.F:362-461: row_p / col_i / blk_p and
one data area), 0-based, with every array living in HBM as a torch tensor.
torch is only the allocator here; all arithmetic goes through the C-ABI."""
import ctypes as C
import itertools
import weakref

import numpy as np
import torch

from . import lib as _lib


def _dtype_code(dt):
    if dt == torch.float64:
        return _lib.dbcsr_type_real_8
    if dt == torch.float32:
        return _lib.dbcsr_type_real_4
    raise TypeError("dbcsr_amd supports real_8 and real_4 data, got %r" % (dt,))


class StreamHandle:
    """The C-ABI stream convention: a pointer to a hipStream_t (dbcsr_acc.h)."""

    def __init__(self, torch_stream=None):
        s = torch_stream if torch_stream is not None else torch.cuda.current_stream()
        self._slot = C.c_void_p(s.cuda_stream)
        self.ptr = C.cast(C.pointer(self._slot), C.c_void_p)


_SERIALS = {}   # id(index tensor) -> (weak reference, serial number); an entry goes when its tensor does
_NEXT_SERIAL = itertools.count(1)


def _serial(t):
    """A number that identifies the tensor OBJECT for its lifetime: a tensor that is freed and another one allocated at the same
    address (or with the same id) never share it (unlike data_ptr)."""
    k = id(t)
    ent = _SERIALS.get(k)
    if ent is not None and ent[0]() is t:
        return ent[1]
    s = next(_NEXT_SERIAL)

    def gone(ref, k=k):
        cur = _SERIALS.get(k)
        if cur is not None and cur[0] is ref:
            del _SERIALS[k]

    _SERIALS[k] = (weakref.ref(t, gone), s)
    return s


class DbcsrMatrix:
    def __init__(self, row_blk_size, col_blk_size, row_p, col_i, blk_p, data, name="", symmetry="N"):
        self._generation = 0   # bumped whenever the library is handed this matrix as a destination (it writes the index arrays)
        self.row_blk_size, self.col_blk_size = row_blk_size, col_blk_size
        self.row_p, self.col_i, self.blk_p, self.data = row_p, col_i, blk_p, data
        self.name = name
        # matrix_type of the reference (src/core/dbcsr_types.F): 'N' no symmetry, 'S' symmetric, 'A' antisymmetric -- the latter
        # two store one block per symmetric pair
        self.symmetry = symmetry

    # -- construction -------------------------------------------------------
    @classmethod
    def from_host(cls, row_blk_size, col_blk_size, row_p, col_i, blk_p, data, device="cuda", name=""):
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(device)
        data = np.ascontiguousarray(data)
        return cls(t(row_blk_size, torch.int32), t(col_blk_size, torch.int32), t(row_p, torch.int32), t(col_i, torch.int32),
                   t(blk_p, torch.int64), torch.as_tensor(data).to(device), name)

    @classmethod
    def empty_like_pattern(cls, row_blk_size, col_blk_size, dtype, device="cuda", name=""):
        nbr = int(row_blk_size.numel())
        return cls(row_blk_size, col_blk_size, torch.zeros(nbr + 1, dtype=torch.int32, device=device),
                   torch.zeros(0, dtype=torch.int32, device=device), torch.zeros(0, dtype=torch.int64, device=device),
                   torch.zeros(0, dtype=dtype, device=device), name)

    # -- properties ---------------------------------------------------------
    @property
    def nblkrows(self):
        return int(self.row_blk_size.numel())

    @property
    def nblkcols(self):
        return int(self.col_blk_size.numel())

    @property
    def nblks(self):
        return int(self.col_i.numel())

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def dtype_code(self):
        return _dtype_code(self.data.dtype)

    def index_stamp(self):
        """dbcsr_amd_bcsr.index_stamp: changes whenever one of the five index tensors is another object, has been written in place by
        torch (its version counter) or by the library (desc(out=True)); never 0."""
        h = 1469598103934665603
        for t in (self.row_blk_size, self.col_blk_size, self.row_p, self.col_i, self.blk_p):
            for v in (_serial(t), t._version):
                h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        h = ((h ^ self._generation) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h or 1

    def desc(self, out=False):
        """The C struct.  out=True: the library is about to write this matrix's index arrays (a destination)."""
        if out:
            self._generation += 1
        p = lambda t: t.data_ptr() if t is not None and t.numel() > 0 else None
        return _lib.BcsrDesc(self.nblkrows, self.nblkcols, p(self.row_blk_size), p(self.col_blk_size), self.row_p.data_ptr(),
                             p(self.col_i), p(self.blk_p), p(self.data), self.nblks, self.index_stamp())

    def to_host(self):
        """(row_blk_size, col_blk_size, row_p, col_i, blk_p, data) as numpy arrays."""
        g = lambda t: t.detach().cpu().numpy()
        return g(self.row_blk_size), g(self.col_blk_size), g(self.row_p), g(self.col_i), g(self.blk_p), g(self.data)

    def copy(self):
        return DbcsrMatrix(self.row_blk_size, self.col_blk_size, self.row_p.clone(), self.col_i.clone(), self.blk_p.clone(),
                           self.data.clone(), self.name)
