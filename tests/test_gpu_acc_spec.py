"""The reference's own C-ABI specification program (tests/dbcsr_acc_test.c: "can serve as a
specification for other backends") compiled from the reference checkout against THIS library by
oracle/build_ref.sh, plus a Python port of the same expectations through ctypes."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from dbcsr_amd import lib as L

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "dbcsr_acc_test")


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/dbcsr_acc_test not built")
def test_reference_spec_program_passes():
    env = dict(os.environ, OMP_NUM_THREADS="8")
    r = subprocess.run([BIN, "0", "8"], env=env, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]


FBIN = os.path.join(os.path.dirname(BIN), "fortran_host_check")


@pytest.mark.skipif(not os.path.exists(FBIN), reason="oracle/_ref/fortran_host_check not built (needs amdflang + /root/reference)")
def test_fortran_host_through_iso_c_binding():
    # a Fortran program (amdflang) using the reference's own dbcsr_acc_device module and ISO_C_BINDING interfaces of the
    # acc ABI: libsmm_acc_transpose + libsmm_acc_process on a 23x23x23 stack, checked against MATMUL
    r = subprocess.run([FBIN], capture_output=True, timeout=120)
    assert r.returncode == 0, (r.stdout.decode()[-500:], r.stderr.decode()[-1500:])
    assert b"fortran host check" in r.stdout


def test_acc_interface_expectations():
    lib = L.load_library()
    n = C.c_int(0)
    assert lib.c_dbcsr_acc_get_ndevices(C.byref(n)) == 0 and n.value >= 1  # legal before init
    assert lib.c_dbcsr_acc_set_active_device(0) == 0
    assert lib.c_dbcsr_acc_init() == 0
    assert lib.libsmm_acc_init() == 0
    free, total = C.c_size_t(), C.c_size_t()
    assert lib.c_dbcsr_acc_dev_mem_info(C.byref(free), C.byref(total)) == 0 and free.value <= total.value
    lo, hi = C.c_int(), C.c_int()
    assert lib.c_dbcsr_acc_stream_priority_range(C.byref(lo), C.byref(hi)) == 0
    s = C.c_void_p()
    for name, prio in ((None, lo.value), (b"", (lo.value + hi.value) // 2), (b"stream", hi.value)):
        assert lib.c_dbcsr_acc_stream_create(C.byref(s), name, prio) == 0
        if name != b"stream":
            assert lib.c_dbcsr_acc_stream_destroy(s) == 0
    ev = C.c_void_p()
    assert lib.c_dbcsr_acc_event_create(C.byref(ev)) == 0
    occurred = C.c_int(0)
    assert lib.c_dbcsr_acc_event_query(ev, C.byref(occurred)) == 0 and occurred.value == 1  # never recorded -> occurred
    nbytes = 16 << 20
    host, dev, view = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.c_dbcsr_acc_host_mem_allocate(C.byref(host), nbytes, s) == 0
    assert lib.c_dbcsr_acc_dev_mem_allocate(C.byref(dev), nbytes) == 0
    C.memset(host, 0xFF, nbytes)
    assert lib.c_dbcsr_acc_memset_zero(dev, 0, nbytes // 2, s) == 0
    assert lib.c_dbcsr_acc_memset_zero(dev, nbytes // 2, nbytes - nbytes // 2, s) == 0
    assert lib.c_dbcsr_acc_memcpy_d2h(dev, host, nbytes, s) == 0
    assert lib.c_dbcsr_acc_event_record(ev, s) == 0
    assert lib.c_dbcsr_acc_stream_wait_event(s, ev) == 0
    assert lib.c_dbcsr_acc_event_synchronize(ev) == 0
    assert lib.c_dbcsr_acc_event_query(ev, C.byref(occurred)) == 0 and occurred.value == 1
    buf = np.frombuffer((C.c_char * nbytes).from_address(host.value), np.uint8)
    assert not buf.any()
    # h2d / d2d / set_ptr view round trip
    buf[:] = np.arange(nbytes, dtype=np.uint64).astype(np.uint8)
    assert lib.c_dbcsr_acc_memcpy_h2d(host, dev, nbytes, s) == 0
    assert lib.c_dbcsr_acc_dev_mem_set_ptr(C.byref(view), dev, 4096) == 0 and view.value == dev.value + 4096
    assert lib.c_dbcsr_acc_memcpy_d2d(view, dev, 4096, s) == 0
    buf2 = np.zeros(4096, np.uint8)
    assert lib.c_dbcsr_acc_memcpy_d2h(dev, buf2.ctypes.data_as(C.c_void_p), 4096, s) == 0
    assert lib.c_dbcsr_acc_stream_sync(s) == 0
    assert np.array_equal(buf2, buf[4096:8192])
    assert lib.c_dbcsr_acc_device_synchronize() == 0
    assert lib.c_dbcsr_acc_event_destroy(ev) == 0
    assert lib.c_dbcsr_acc_dev_mem_deallocate(dev) == 0
    assert lib.c_dbcsr_acc_host_mem_deallocate(host, s) == 0
    assert lib.c_dbcsr_acc_stream_destroy(s) == 0
    lib.c_dbcsr_acc_clear_errors()
    assert lib.libsmm_acc_finalize() == 0
    assert lib.c_dbcsr_acc_finalize() == 0
