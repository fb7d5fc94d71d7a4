"""Device plumbing: PyTorch-ROCm tensors are the device buffers handed to the C ABI (pointers + stream)."""
import ctypes

import numpy as np
import torch

from . import _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("pta_replicator_amd needs an AMD GPU visible to PyTorch-ROCm (torch.cuda.is_available() "
                           "is False); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def f64(x, device=None):
    """contiguous float64 device tensor from array-like (copy)."""
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float64)), device=device or require_gpu())


def i32(x, device=None):
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.int32)), device=device or require_gpu())


def i64(x, device=None):
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.int64)), device=device or require_gpu())


def empty(shape, dtype=torch.float64, device=None):
    return torch.empty(shape, dtype=dtype, device=device or require_gpu())


def zeros(shape, dtype=torch.float64, device=None):
    return torch.zeros(shape, dtype=dtype, device=device or require_gpu())


def ptr(t):
    """raw device pointer of a tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def hptr(a):
    """raw host pointer of a contiguous numpy array."""
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def device_info():
    cu, wf = ctypes.c_int(0), ctypes.c_int(0)
    arch = ctypes.create_string_buffer(64)
    _lib.call("pta_device_info", ctypes.byref(cu), ctypes.byref(wf), arch, 64)
    return {"cu_count": cu.value, "wavefront": wf.value, "arch": arch.value.decode()}
