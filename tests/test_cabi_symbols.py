"""CPU-side boundary checks: the C-ABI library builds, loads, and exports every
function that include/*.h declares (no compute calls: there is no GPU here)."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        txt = open(h).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        for mm in re.finditer(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_\s\*]*?\b([a-z_][A-Za-z0-9_]*)\s*\(", txt, flags=re.M):
            nm = mm.group(1)
            if nm in ("defined", "if", "sizeof"):
                continue
            names.append(nm)
    return sorted(set(names))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as g
    g.build_native()
    from dbcsr_amd import lib
    return lib.load_library()


def test_header_declares_reference_abi():
    from dbcsr_amd import lib
    names = declared_functions()
    for s in lib.ACC_SYMBOLS + lib.LIBSMM_SYMBOLS + lib.MM_SYMBOLS + lib.COMM_SYMBOLS:
        assert s in names, s
    # the 26 acc.h functions + 2 imported timing hooks, 6 (+1) libsmm functions
    assert len(lib.ACC_SYMBOLS) == 28 and len(lib.LIBSMM_SYMBOLS) == 7


def test_library_exports_every_declared_symbol(native):
    for nm in declared_functions():
        assert hasattr(native, nm), "libdbcsr_acc_amd.so does not export %s" % nm


def test_thread_safe_and_warp_size(native):
    # no device needed: pure host answers (src/core/dbcsr_lib.F:248-262 handshake)
    assert native.libsmm_acc_is_thread_safe() == 1
    assert native.libsmm_acc_gpu_warp_size() == 64


def test_get_ndevices_before_init_without_gpu(native):
    n = ctypes.c_int(-1)
    assert native.c_dbcsr_acc_get_ndevices(ctypes.byref(n)) == 0
    assert n.value >= 0
    assert native.c_dbcsr_acc_dev_mem_deallocate(None) == 0  # NULL is legal (dbcsr_acc_test.c:189)
    assert native.c_dbcsr_acc_stream_destroy(None) == 0
    assert native.c_dbcsr_acc_event_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dbcsr_amd import lib
    monkeypatch.setattr(lib, "_LIB", None)
    monkeypatch.setattr(lib, "_LIB_LAB", None)
    monkeypatch.setattr(lib, "library_path", lambda lab=False: str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load_library()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load_library(lab=True)
