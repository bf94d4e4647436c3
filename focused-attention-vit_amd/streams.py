"""HIP streams restricted to a subset of the compute units.

The segmentation of batch k + 1 (SLIC: grids that fill every CU for 65 us at a time) runs under the training step of
batch k, which at the SPPP token counts is a chain of ~240 short launches.  On an ordinary side stream the two time-share
the GPU: every short launch of the step queues behind the segmentation's workgroups, and the pair takes the SUM of
their times.  A stream created with `hipExtStreamCreateWithCUMask` keeps the segmentation on N of
the 256 CUs and leaves the others idle for the step's launches.  (Reference: the SLIC call it moves is
`/root/reference/models/sppp.py:63-66`, executed on the CPU inside `forward` there.)
"""
import ctypes as C

import torch

_HIP = None
_LIVE = []          # (stream handle, ExternalStream): never destroyed while the process lives -- graphs may reference them


def _hip():
    global _HIP
    if _HIP is None:
        _HIP = C.CDLL("libamdhip64.so")      # the runtime torch already loaded (same soname)
        _HIP.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
        _HIP.hipExtStreamCreateWithCUMask.restype = C.c_int
    return _HIP


def cu_masked_stream(n_cus: int, device=None) -> "torch.cuda.Stream":
    """A stream whose kernels run on the first `n_cus` bits of the device's CU mask (the driver deals the bits round
    robin over the XCDs and shader engines, so any prefix is spread evenly).  `n_cus` >= the CU count: a plain stream."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    if n_cus <= 0:
        raise ValueError("cu_masked_stream: n_cus must be positive")
    if n_cus >= total:
        return torch.cuda.Stream(device=dev)
    words = (total + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(n_cus):
        mask[i // 32] |= 1 << (i % 32)
    handle = C.c_void_p()
    with torch.cuda.device(dev):
        rc = _hip().hipExtStreamCreateWithCUMask(C.byref(handle), words, mask)
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask({n_cus} of {total} CUs) failed: hipError {rc}")
    s = torch.cuda.ExternalStream(handle.value, device=dev)
    _LIVE.append((handle, s))
    return s
