"""dbcsr_amd -- MI355X-native block-sparse multiply behind DBCSR's interfaces.

Host side (Python) above the C-ABI shared library ``libdbcsr_acc_amd.so``
(HIP kernels for gfx950).  PyTorch is used for device memory, streams and
``torch.distributed`` only.  There is no CPU fallback: importing
:mod:`dbcsr_amd.lib` raises if the native library is missing.
"""
from .lib import load_library, library_path  # noqa: F401

__all__ = ["load_library", "library_path"]
