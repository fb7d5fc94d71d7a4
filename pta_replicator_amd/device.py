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


# ---- staging for the drop-in API: ONE pinned host buffer per direction, grown on demand and reused across calls, so that an add_*
# call is one host -> device copy, its kernels and one device -> host copy (a pageable copy of every operand vector through the
# driver's own bounce buffers, each with its own synchronisation, was most of a call's latency)
_pinned = {}


def _pinned_buffer(kind, nbytes):
    buf = _pinned.get(kind)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty((max(int(nbytes), 1 << 20),), dtype=torch.uint8).pin_memory()
        _pinned[kind] = buf
    return buf


def upload_packed(arrays):
    """float64 / int32 host arrays -> device tensors through one pinned staging copy.  Returns one device tensor per input (views
    of a single device buffer, 16-byte aligned each)."""
    require_gpu()
    metas, off = [], 0
    for a in arrays:
        a = np.ascontiguousarray(a)
        if a.dtype not in (np.float64, np.int32):
            a = a.astype(np.float64)
        metas.append((a, off))
        off += (a.nbytes + 15) // 16 * 16
    ev = _pinned.get("h2d_event")
    if ev is not None:
        ev.synchronize()          # the previous upload must have left the staging buffer before it is overwritten
    host = _pinned_buffer("h2d", off)
    hv = host.numpy()
    for a, o in metas:
        hv[o:o + a.nbytes] = a.view(np.uint8).reshape(-1)
    dev = host[:off].to(require_gpu(), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    _pinned["h2d_event"] = ev
    out = []
    for a, o in metas:
        t = dev[o:o + a.nbytes].view(torch.float64 if a.dtype == np.float64 else torch.int32)
        out.append(t.view(a.shape))
    return out


def download(t):
    """device tensor -> NumPy array (a fresh copy) through the pinned download buffer: one asynchronous copy + one synchronisation."""
    nbytes = t.numel() * t.element_size()
    host = _pinned_buffer("d2h", nbytes)
    hv = host[:nbytes].view(t.dtype).view(t.shape)
    hv.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return hv.numpy().copy()
