"""PyTorch-ROCm plumbing: device buffers, raw pointers and the current HIP stream.

torch is used for memory and streams only; every kernel is in libdpilqr_hip.so.
"""
import numpy as np
import torch

from . import _lib


def device():
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def stream_handle():
    """hipStream_t of torch's current stream, as an int for ctypes (void*)."""
    return torch.cuda.current_stream().cuda_stream


def to_dev(a, dtype=torch.float64):
    """Host array -> contiguous device tensor."""
    if isinstance(a, torch.Tensor):
        return a.to(device=device(), dtype=dtype).contiguous()
    np_dtype = {torch.float64: np.float64, torch.float32: np.float32}.get(dtype, np.int32)
    return torch.from_numpy(np.array(a, dtype=np_dtype, order="C", copy=True)).to(device())


def empty(shape, dtype=torch.float64):
    return torch.empty(shape, dtype=dtype, device=device())


def zeros(shape, dtype=torch.float64):
    return torch.zeros(shape, dtype=dtype, device=device())


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("device tensor must be contiguous")
    return t.data_ptr()
