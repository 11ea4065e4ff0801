"""ctypes loader of libdclnet_hip.so (the C ABI declared in include/dclnet_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails, a RuntimeError
is raised.  PyTorch is used for device memory and streams only.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libdclnet_hip.so")
DIAG_SO_PATH = os.path.join(os.path.dirname(_HERE), "tests", "_diag", "libdclnet_hip_diag.so")
_LIB = None

vp = C.c_void_p


def build(verbose=False, diag=False):
    """hipcc --offload-arch=gfx950 build of csrc/ into libdclnet_hip.so (in-tree); diag=True also builds the diagnostic
    library (tests/_diag/libdclnet_hip_diag.so: the same sources with -DDCL_DIAG, see csrc/Makefile)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"] + (["all", "diag"] if diag else []), stdout=out)
    return SO_PATH


ABI_VERSION = 2                 # include/dclnet_hip.h: DCL_ABI_VERSION this package's ctypes calls are written against


def _open(path):
    L = C.CDLL(path)
    L.dcl_last_error.restype = C.c_char_p
    got = int(L.dcl_abi_version())
    if got != ABI_VERSION:
        raise RuntimeError("%s speaks C-ABI version %d, this package calls version %d (include/dclnet_hip.h): rebuild the "
                           "library (make -C dcl-net_amd/csrc)" % (path, got, ABI_VERSION))
    return L


def lib():
    """the product library.  It has no tuning hooks and reads no environment; there is no CPU fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "libdclnet_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dcl-net_amd/csrc`.  There is no CPU fallback." % SO_PATH)
        _LIB = _open(SO_PATH)
    return _LIB


class diagnostic_library(object):
    """`with _native.diagnostic_library() as L:` -- tests/ and tools/ only.  Inside the block every op of this package
    calls into the DIAGNOSTIC build (tests/_diag/libdclnet_hip_diag.so or `path`: same sources, -DDCL_DIAG) whose
    dcl_debug_* hooks select kernel variants; the product library is restored on exit."""

    def __init__(self, path=None):
        self.path = path or DIAG_SO_PATH

    def __enter__(self):
        global _LIB
        if not os.path.exists(self.path):
            raise RuntimeError("diagnostic library missing (%s): make -C dcl-net_amd/csrc diag" % self.path)
        self.prev = _LIB
        _LIB = _open(self.path)
        return _LIB

    def __exit__(self, *exc):
        global _LIB
        _LIB = self.prev
        return False


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("libdclnet_hip %s failed (rc=%d): %s" % (what, rc, lib().dcl_last_error().decode()))


def stream():
    return vp(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device/host pointer of a tensor (None -> NULL)."""
    if t is None:
        return vp(0)
    return vp(t.data_ptr())


def need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dcl-net_amd: this op runs on the GPU only (got a %s tensor); no CPU fallback" % t.device)


def f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def i32c(t):
    return t.contiguous() if t.dtype == torch.int32 else t.int().contiguous()
