"""ctypes binding of include/polar_hip_testing.h -- the TEST-ONLY entry points (kernel selection for cross-checks, the
kernels' scalar arithmetic on chosen operands).  They live in libpolar_hip_testing.so, a second build of the same sources
with -DPOLAR_TESTING; the product library libpolar_hip.so does not export them.  A decoder handed to select_kernel /
big_split is re-created inside the test library first.  Used by tests/ and tools/, never by the product."""
import ctypes as C

import numpy as np

from .api import PolarError, load_library

KERNEL_AUTO, KERNEL_GENERIC, KERNEL_GENERIC_SPILL, KERNEL_BIG, KERNEL_ONE_PER_WAVE, KERNEL_FOUR_PER_WAVE = 0, 1, 2, 3, 4, 5
OP_CHK, OP_CHK_LUT, OP_CHK_LUT1, OP_TABV, OP_PHI, OP_PHI_LUT, OP_CHK_CNT, OP_CHK_IDX, OP_CHK_TAB = 0, 1, 2, 3, 4, 5, 6, 7, 8


def select_kernel(dec, variant):
    """dec: a polardecoding_amd Decoder.  Returns dec (its kernel_name reflects the choice)."""
    lib = load_library(testing=True)
    dec._rebind(lib)
    lib.polar_testing_select_kernel.argtypes = [C.c_void_p, C.c_int]
    rc = lib.polar_testing_select_kernel(dec._h, int(variant))
    if rc != 0:
        raise PolarError(f"polar_testing_select_kernel rc={rc}")
    return dec


def big_split(dec, split):
    lib = load_library(testing=True)
    dec._rebind(lib)
    lib.polar_testing_big_split.argtypes = [C.c_void_p, C.c_int]
    rc = lib.polar_testing_big_split(dec._h, int(split))
    if rc != 0:
        raise PolarError(f"polar_testing_big_split rc={rc}")
    return dec


def math(op, a, b, dtype=np.float64, device=0):
    """The device functions of csrc/polar_math.h / polar_lut.h applied element-wise on the GPU."""
    lib = load_library(testing=True)
    lib.polar_testing_math.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    a = np.ascontiguousarray(a, dtype=dtype).ravel()
    b = np.ascontiguousarray(np.broadcast_to(np.asarray(b, dtype=dtype), a.shape), dtype=dtype).ravel()
    out = np.empty_like(a)
    rc = lib.polar_testing_math(int(op), 1 if dtype == np.float32 else 0, a.ctypes.data, b.ctypes.data,
                                out.ctypes.data, a.size, device)
    if rc != 0:
        raise PolarError(f"polar_testing_math rc={rc}")
    return out
