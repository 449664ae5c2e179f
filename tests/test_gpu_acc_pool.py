"""The caching allocator behind c_dbcsr_acc_{dev,host}_mem_{allocate,deallocate} (csrc/acc_runtime.hip): released blocks are
handed out again (the host's C buffer grows and is released in every multiply: dbcsr_mm_accdrv.F:475-476), contents written
through one allocation are not visible through a neighbour, the blocks the library keeps count as free device memory."""
import ctypes as C

import numpy as np
import pytest

from dbcsr_amd import lib as _lib

pytestmark = pytest.mark.gpu


def test_released_blocks_are_reused_and_counted_free():
    L = _lib.load_library()
    assert L.c_dbcsr_acc_init() == 0
    f0, t0, f1 = C.c_size_t(), C.c_size_t(), C.c_size_t()
    n = 192 << 20
    p = C.c_void_p()
    assert L.c_dbcsr_acc_dev_mem_allocate(C.byref(p), n) == 0 and p.value
    first = p.value
    assert L.c_dbcsr_acc_dev_mem_info(C.byref(f0), C.byref(t0)) == 0
    assert L.c_dbcsr_acc_dev_mem_deallocate(p) == 0
    assert L.c_dbcsr_acc_dev_mem_info(C.byref(f1), C.byref(t0)) == 0
    assert f1.value >= f0.value + n - (1 << 20)          # the cached block is available to the host
    q = C.c_void_p()
    assert L.c_dbcsr_acc_dev_mem_allocate(C.byref(q), n - 4096) == 0
    assert q.value == first                               # same size class: the released block comes back
    small = C.c_void_p()
    assert L.c_dbcsr_acc_dev_mem_allocate(C.byref(small), 1 << 20) == 0 and small.value != first
    assert L.c_dbcsr_acc_dev_mem_deallocate(q) == 0 and L.c_dbcsr_acc_dev_mem_deallocate(small) == 0
    # pinned host memory: same behaviour, and the memory is usable
    h = C.c_void_p()
    assert L.c_dbcsr_acc_host_mem_allocate(C.byref(h), 8 << 20, None) == 0 and h.value
    hfirst = h.value
    a = np.ctypeslib.as_array(C.cast(h, C.POINTER(C.c_double)), shape=(1 << 20,))
    a[:] = 3.5
    assert L.c_dbcsr_acc_host_mem_deallocate(h, None) == 0
    h2 = C.c_void_p()
    assert L.c_dbcsr_acc_host_mem_allocate(C.byref(h2), 8 << 20, None) == 0 and h2.value == hfirst
    assert L.c_dbcsr_acc_host_mem_deallocate(h2, None) == 0
    # a few hundred allocate / release rounds of growing sizes (the C buffer's life in a multiply) stay within the device
    for rnd in range(3):
        size = 1 << 20
        held = []
        while size < (1 << 30):
            b = C.c_void_p()
            assert L.c_dbcsr_acc_dev_mem_allocate(C.byref(b), size) == 0
            held.append(b)
            if len(held) > 1:
                assert L.c_dbcsr_acc_dev_mem_deallocate(held.pop(0)) == 0
            size = int(size * 1.6)
        for b in held:
            assert L.c_dbcsr_acc_dev_mem_deallocate(b) == 0
    assert L.c_dbcsr_acc_finalize() == 0                  # empties the pools
    assert L.c_dbcsr_acc_init() == 0
