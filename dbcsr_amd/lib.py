"""ctypes binding of libdbcsr_acc_amd.so (include/*.h).  Fails loudly when the
library has not been built -- there is no fallback path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_LAB = None

# DBCSR data type codes (reference src/data/dbcsr_data_types.F:122-133)
dbcsr_type_real_4 = 1
dbcsr_type_real_8 = 3
dbcsr_type_complex_4 = 5
dbcsr_type_complex_8 = 7

ACC_SYMBOLS = [
    "c_dbcsr_acc_init", "c_dbcsr_acc_finalize", "c_dbcsr_acc_clear_errors", "c_dbcsr_acc_get_ndevices",
    "c_dbcsr_acc_set_active_device", "c_dbcsr_acc_device_synchronize", "c_dbcsr_acc_stream_priority_range",
    "c_dbcsr_acc_stream_create", "c_dbcsr_acc_stream_destroy", "c_dbcsr_acc_stream_sync", "c_dbcsr_acc_stream_wait_event",
    "c_dbcsr_acc_event_create", "c_dbcsr_acc_event_destroy", "c_dbcsr_acc_event_record", "c_dbcsr_acc_event_query",
    "c_dbcsr_acc_event_synchronize", "c_dbcsr_acc_dev_mem_allocate", "c_dbcsr_acc_dev_mem_deallocate",
    "c_dbcsr_acc_dev_mem_set_ptr", "c_dbcsr_acc_host_mem_allocate", "c_dbcsr_acc_host_mem_deallocate",
    "c_dbcsr_acc_memcpy_h2d", "c_dbcsr_acc_memcpy_d2h", "c_dbcsr_acc_memcpy_d2d", "c_dbcsr_acc_memset_zero",
    "c_dbcsr_acc_dev_mem_info", "c_dbcsr_timeset", "c_dbcsr_timestop",
]
LIBSMM_SYMBOLS = [
    "libsmm_acc_init", "libsmm_acc_finalize", "libsmm_acc_is_thread_safe", "libsmm_acc_transpose", "libsmm_acc_process",
    "c_calculate_norms", "libsmm_acc_gpu_warp_size",
]
MM_SYMBOLS = [
    "dbcsr_amd_mm_create", "dbcsr_amd_mm_destroy", "dbcsr_amd_mm_symbolic", "dbcsr_amd_mm_numeric", "dbcsr_amd_bcsr_transpose",
    "dbcsr_amd_bcsr_checksum", "dbcsr_amd_bcsr_fill_random", "dbcsr_amd_mm_kernel_name", "dbcsr_amd_mm_last_kernel", "dbcsr_amd_fabric_probe", "dbcsr_amd_mm_plan_stats", "dbcsr_amd_mm_trust_plan", "dbcsr_amd_mm_expect_filter", "dbcsr_amd_mm_stats", "dbcsr_amd_mm_timing", "dbcsr_amd_mm_init_c", "dbcsr_amd_bcsr_fill_random_dist",
    "dbcsr_amd_mm_symbolic_filtered", "dbcsr_amd_bcsr_filter_count", "dbcsr_amd_bcsr_filter_apply",
    "dbcsr_amd_bcsr_crop_count", "dbcsr_amd_bcsr_crop_apply", "dbcsr_amd_bcsr_scale_window",
    "dbcsr_amd_multiply", "dbcsr_amd_bcsr_release", "dbcsr_amd_bcsr_desymmetrize_count", "dbcsr_amd_bcsr_desymmetrize_apply",
    "dbcsr_amd_bcsr_twin_count", "dbcsr_amd_bcsr_twin_apply", "dbcsr_amd_mm_set_canonical_product", "dbcsr_amd_multiply_symmetric_c",
    "dbcsr_amd_bcsr_desymmetrized", "dbcsr_amd_smm_last_kernel", "dbcsr_amd_multiply_symmetric_c_klimits",
]


COMM_SYMBOLS = ["dbcsr_amd_comm_available", "dbcsr_amd_comm_unique_id", "dbcsr_amd_comm_create", "dbcsr_amd_comm_destroy", "dbcsr_amd_comm_rank", "dbcsr_amd_comm_exchange",
                "dbcsr_amd_comm_allgather"]


class CommOp(C.Structure):
    """struct dbcsr_amd_comm_op (include/dbcsr_amd_comm.h)"""
    _fields_ = [("buf", C.c_void_p), ("bytes", C.c_int64), ("peer", C.c_int32), ("reserved", C.c_int32)]


class BcsrDesc(C.Structure):
    """struct dbcsr_amd_bcsr (include/dbcsr_amd_mm.h); device pointers."""
    _fields_ = [("nblkrows", C.c_int32), ("nblkcols", C.c_int32), ("row_blk_size", C.c_void_p), ("col_blk_size", C.c_void_p),
                ("row_p", C.c_void_p), ("col_i", C.c_void_p), ("blk_p", C.c_void_p), ("data", C.c_void_p), ("nblks", C.c_int64), ("index_stamp", C.c_uint64)]


class MnkStat(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("reserved", C.c_int32), ("nproducts", C.c_int64), ("flop", C.c_int64)]


class MmCounts(C.Structure):
    _fields_ = [("c_nblks", C.c_int64), ("c_nze", C.c_int64), ("nproducts", C.c_int64), ("flop", C.c_int64)]


def library_path(lab=False):
    """The shipping build, or the lab build (dbcsr_amd/csrc/Makefile: the same library plus the dataflows and variants that were built,
    parity-tested and measured but do not win -- what the DBCSR_AMD_MM_TILE / _BAND / _HOT_VARIANT ... switches select)."""
    return os.path.join(_HERE, "libdbcsr_acc_amd_lab.so" if lab else "libdbcsr_acc_amd.so")


def want_lab():
    """DBCSR_AMD_LAB=1: engines made from now on use the lab build (the tests of the experimental kernels, profiling sessions)."""
    return os.environ.get("DBCSR_AMD_LAB", "0") not in ("", "0")


def load_library(lab=False):
    global _LIB, _LIB_LAB
    if lab and _LIB_LAB is not None:
        return _LIB_LAB
    if not lab and _LIB is not None:
        return _LIB
    path = library_path(lab)
    # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  The HIP runtime that is loaded FIRST wins
    # (same SONAME); if this library were loaded before torch, two different runtimes could end up in one process and
    # device pointers allocated by torch would be foreign to the one this library talks to.  So when torch is
    # installed, make sure it is loaded first.  (Pure C/Fortran hosts link the library against /opt/rocm directly.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise RuntimeError(
            "dbcsr_amd: native library %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C dbcsr_amd/csrc`. There is no CPU fallback." % path)
    L = C.CDLL(path)
    vp, i32, i64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
    L.c_dbcsr_acc_init.restype = i32
    L.c_dbcsr_acc_finalize.restype = i32
    L.c_dbcsr_acc_clear_errors.restype = None
    L.c_dbcsr_acc_get_ndevices.argtypes = [C.POINTER(i32)]
    L.c_dbcsr_acc_set_active_device.argtypes = [i32]
    L.c_dbcsr_acc_stream_priority_range.argtypes = [C.POINTER(i32), C.POINTER(i32)]
    L.c_dbcsr_acc_stream_create.argtypes = [C.POINTER(vp), C.c_char_p, i32]
    L.c_dbcsr_acc_stream_destroy.argtypes = [vp]
    L.c_dbcsr_acc_stream_sync.argtypes = [vp]
    L.c_dbcsr_acc_stream_wait_event.argtypes = [vp, vp]
    L.c_dbcsr_acc_event_create.argtypes = [C.POINTER(vp)]
    L.c_dbcsr_acc_event_destroy.argtypes = [vp]
    L.c_dbcsr_acc_event_record.argtypes = [vp, vp]
    L.c_dbcsr_acc_event_query.argtypes = [vp, C.POINTER(i32)]
    L.c_dbcsr_acc_event_synchronize.argtypes = [vp]
    L.c_dbcsr_acc_dev_mem_allocate.argtypes = [C.POINTER(vp), sz]
    L.c_dbcsr_acc_dev_mem_deallocate.argtypes = [vp]
    L.c_dbcsr_acc_dev_mem_set_ptr.argtypes = [C.POINTER(vp), vp, sz]
    L.c_dbcsr_acc_host_mem_allocate.argtypes = [C.POINTER(vp), sz, vp]
    L.c_dbcsr_acc_host_mem_deallocate.argtypes = [vp, vp]
    L.c_dbcsr_acc_memcpy_h2d.argtypes = [vp, vp, sz, vp]
    L.c_dbcsr_acc_memcpy_d2h.argtypes = [vp, vp, sz, vp]
    L.c_dbcsr_acc_memcpy_d2d.argtypes = [vp, vp, sz, vp]
    L.c_dbcsr_acc_memset_zero.argtypes = [vp, sz, sz, vp]
    L.c_dbcsr_acc_dev_mem_info.argtypes = [C.POINTER(sz), C.POINTER(sz)]
    L.libsmm_acc_transpose.argtypes = [vp, i32, i32, vp, i32, i32, i32, i32, vp]
    L.libsmm_acc_process.argtypes = [vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
    L.c_calculate_norms.argtypes = [vp, i32, vp, vp, vp, vp]
    L.dbcsr_amd_mm_create.argtypes = [C.POINTER(vp)]
    L.dbcsr_amd_mm_destroy.argtypes = [vp]
    BP = C.POINTER(BcsrDesc)
    L.dbcsr_amd_mm_symbolic.argtypes = [vp, BP, BP, BP, i32, vp, C.POINTER(MmCounts), vp]
    L.dbcsr_amd_mm_numeric.argtypes = [vp, i32, C.c_double, BP, BP, C.c_double, BP, BP, vp]
    L.dbcsr_amd_bcsr_transpose.argtypes = [vp, i32, BP, BP, vp]
    L.dbcsr_amd_bcsr_checksum.argtypes = [vp, i32, BP, C.POINTER(C.c_double), vp]
    L.dbcsr_amd_bcsr_fill_random.argtypes = [vp, i32, BP, i32, vp]
    L.dbcsr_amd_mm_symbolic_filtered.argtypes = [vp, i32, C.c_double, C.c_double, BP, BP, BP, i32, vp, C.POINTER(MmCounts), vp]
    L.dbcsr_amd_bcsr_filter_count.argtypes = [vp, i32, BP, C.c_double, vp, C.POINTER(i64), C.POINTER(i64), vp]
    L.dbcsr_amd_bcsr_filter_apply.argtypes = [vp, i32, BP, BP, vp]
    L.dbcsr_amd_bcsr_crop_count.argtypes = [vp, i32, BP, i64, i64, i64, i64, vp, C.POINTER(i64), C.POINTER(i64), vp]
    L.dbcsr_amd_bcsr_crop_apply.argtypes = [vp, i32, BP, BP, vp]
    L.dbcsr_amd_bcsr_scale_window.argtypes = [vp, i32, BP, C.c_double, i64, i64, i64, i64, vp]
    L.dbcsr_amd_multiply.argtypes = [vp, C.c_char, C.c_char, i32, C.c_double, BP, BP, C.c_double, BP, C.POINTER(i64), i32, C.c_double, BP,
                                     C.POINTER(i64), vp]
    L.dbcsr_amd_bcsr_release.argtypes = [BP]
    L.dbcsr_amd_mm_init_c.argtypes = [vp, i32, C.c_double, BP, BP, vp]
    L.dbcsr_amd_bcsr_fill_random_dist.argtypes = [vp, i32, BP, i32, vp, vp, i32, vp]
    L.dbcsr_amd_mm_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.dbcsr_amd_mm_kernel_name.argtypes = [i32]
    L.dbcsr_amd_mm_kernel_name.restype = C.c_char_p
    L.dbcsr_amd_bcsr_desymmetrize_count.argtypes = [vp, BP, vp, C.POINTER(i64), C.POINTER(i64), vp]
    L.dbcsr_amd_bcsr_desymmetrize_apply.argtypes = [vp, i32, BP, i32, BP, vp]
    L.dbcsr_amd_bcsr_twin_count.argtypes = [vp, BP, i32, vp, C.POINTER(i64), C.POINTER(i64), vp]
    L.dbcsr_amd_bcsr_twin_apply.argtypes = [vp, i32, BP, i32, i32, BP, vp]
    L.dbcsr_amd_mm_set_canonical_product.argtypes = [vp, i32]
    L.dbcsr_amd_bcsr_desymmetrized.argtypes = [vp, i32, BP, i32, BP, vp]
    L.dbcsr_amd_multiply_symmetric_c.argtypes = [vp, C.c_char, C.c_char, i32, C.c_double, BP, BP, C.c_double, BP, i32, i32, C.c_double, BP,
                                                 C.POINTER(i64), vp]
    L.dbcsr_amd_mm_stats.argtypes = [vp, C.POINTER(MnkStat), i32, C.POINTER(i32), vp]
    L.dbcsr_amd_comm_available.argtypes = []
    L.dbcsr_amd_comm_unique_id.argtypes = [C.c_char_p]
    L.dbcsr_amd_comm_create.argtypes = [C.POINTER(vp), C.c_char_p, i32, i32]
    L.dbcsr_amd_comm_destroy.argtypes = [vp]
    L.dbcsr_amd_comm_rank.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.dbcsr_amd_comm_exchange.argtypes = [vp, C.POINTER(CommOp), i32, C.POINTER(CommOp), i32, vp]
    L.dbcsr_amd_comm_allgather.argtypes = [vp, vp, vp, i64, vp]
    L.dbcsr_amd_mm_last_kernel.argtypes = [vp]
    L.dbcsr_amd_mm_last_kernel.restype = C.c_char_p
    L.dbcsr_amd_smm_last_kernel.argtypes = []
    L.dbcsr_amd_smm_last_kernel.restype = C.c_char_p
    L.dbcsr_amd_fabric_probe.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.dbcsr_amd_mm_plan_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.dbcsr_amd_mm_trust_plan.argtypes = [vp, C.c_int]
    L.dbcsr_amd_mm_expect_filter.argtypes = [vp, C.c_double]
    L.dbcsr_amd_mm_expect_filter.restype = C.c_int
    if lab:   # diagnostics of the experimental dataflows (dbcsr_amd/csrc/mm_lab_api.h): the shipping build does not export them
        L.dbcsr_amd_mm_tile_stats.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
        L.dbcsr_amd_mm_band_stats.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
        _LIB_LAB = L
    else:
        _LIB = L
    return L
