"""ctypes loader of libdclnet_hip.so (the C ABI declared in include/dclnet_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails, a RuntimeError
is raised.  PyTorch is used for device memory and streams only.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("DCL_HIP_LIB") or os.path.join(_HERE, "libdclnet_hip.so")   # DCL_HIP_LIB: diagnostic builds (tools/)
_LIB = None

vp = C.c_void_p


def build(verbose=False):
    """hipcc --offload-arch=gfx950 build of csrc/ into libdclnet_hip.so (in-tree)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], stdout=out)
    return SO_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "libdclnet_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dcl-net_amd/csrc`.  There is no CPU fallback." % SO_PATH)
        L = C.CDLL(SO_PATH)
        L.dcl_last_error.restype = C.c_char_p
        # A/B switches for profiling runs (kernel-variant hooks of include/dclnet_hip.h)
        if os.environ.get("DCL_CONV_VARIANT"):
            L.dcl_debug_force_valu_conv(int(os.environ["DCL_CONV_VARIANT"]))
        if os.environ.get("DCL_CONV_XCD"):
            L.dcl_debug_conv_xcd_remap(int(os.environ["DCL_CONV_XCD"]))
        if os.environ.get("DCL_CONV_SPLIT"):
            L.dcl_debug_conv_split(int(os.environ["DCL_CONV_SPLIT"]))
        if os.environ.get("DCL_CONV_SLOTS"):
            L.dcl_debug_conv_slots(int(os.environ["DCL_CONV_SLOTS"]))
        if os.environ.get("DCL_ATTN_SPLIT"):
            L.dcl_debug_attention_split(int(os.environ["DCL_ATTN_SPLIT"]))
        if os.environ.get("DCL_NN_GRID"):
            L.dcl_debug_three_nn_grid(int(os.environ["DCL_NN_GRID"]))
        if os.environ.get("DCL_ATTN_XCD"):
            L.dcl_debug_attention_xcd_remap(int(os.environ["DCL_ATTN_XCD"]))
        if os.environ.get("DCL_ATTN_VARIANT"):
            L.dcl_debug_attention_variant(int(os.environ["DCL_ATTN_VARIANT"]))
        _LIB = L
    return _LIB


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
