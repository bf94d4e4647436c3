#!/usr/bin/env python3
"""hipMemsetAsync captured into a HIP graph, replayed: is the destination zero after EVERY replay?
No kernel of this repository is involved (torch only provides the capture and the buffers).  On ROCm 7.2 / gfx950 the
second and later replays leave dword 2 of every 16 bytes at an arbitrary constant for some sizes; libfavit therefore
zero-fills with a kernel of its own (csrc/common.h: favit_zero_async)."""
import ctypes as C
import sys

import torch

hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemsetAsync.restype = C.c_int
rc = 0
for nfloat in (48, 192, 4096, 12480, 65 * 192 + 192, 1 << 20):
    buf = torch.ones(nfloat, device="cuda")
    other = torch.ones(nfloat, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        hip.hipMemsetAsync(buf.data_ptr(), 0, 4 * nfloat, torch.cuda.current_stream().cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        other.add_(1.0)                                   # some other node in the graph
        hip.hipMemsetAsync(buf.data_ptr(), 0, 4 * nfloat, torch.cuda.current_stream().cuda_stream)
        other.mul_(0.5)
    line = []
    for rep in range(4):
        buf.fill_(1.0)
        g.replay()
        torch.cuda.synchronize()
        nz = (buf != 0).nonzero().flatten()
        if nz.numel():
            rc = 1
            line.append(f"replay {rep}: {nz.numel()} non-zero dwords, index mod 4 in {sorted(set((nz % 4).tolist()))}, "
                        f"value {buf[nz[0]].item():.6g}")
        else:
            line.append(f"replay {rep}: zero")
    print(f"{4 * nfloat:>8} bytes: " + "; ".join(line), flush=True)
sys.exit(rc)
