#!/usr/bin/env python3
"""Do the two co-resident workgroups of a CU share operand lines through the CU's L1 when they stream the SAME A tile
(adjacent N tiles of one M panel)?  tools/probe_kernels.hip: dma_share_probe.  Prints the workgroups-per-CU census and
the aggregate L2/L1 -> LDS rate for distinct vs shared A tiles."""
import ctypes, os, sys
import torch
pl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe.so"))
pl.probe_dma_share.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                               ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda"
for ksteps in (12, 48):
    ntile = 8
    a_tile = 256 * ksteps * 64
    w_tile = 128 * ksteps * 64
    A = torch.randint(0, 255, (4096 * 2 * ntile * a_tile // 64,), device=dev, dtype=torch.uint8) if False else None
    # keys are < 4096 but only 256 are populated: index A by a dense rank would need a second pass; instead size A for
    # the largest key actually seen (census pass first)
    census = torch.zeros(4096, dtype=torch.int32, device=dev)
    W = torch.randint(0, 255, (64 * w_tile,), device=dev, dtype=torch.uint8)
    tiny = torch.zeros(4096 * 2 * ntile * 64 + a_tile, device=dev, dtype=torch.uint8)       # stride-64 dummy for the census pass
    st = torch.cuda.current_stream().cuda_stream
    pl.probe_dma_share(tiny.data_ptr(), W.data_ptr(), 64, w_tile, ntile, 1, 0, census.data_ptr(), st)
    torch.cuda.synchronize()
    c = census.cpu()
    keys = (c > 0).nonzero().flatten()
    print(f"ksteps={ksteps}: {len(keys)} CUs seen, workgroups per CU: " + ", ".join(f"{int(v)}x{int((c == v).sum())}" for v in c[c > 0].unique()))
    maxkey = int(keys.max()) + 1
    A = torch.randint(0, 255, (maxkey * 2 * ntile * a_tile,), device=dev, dtype=torch.uint8)
    for mode, name in ((0, "distinct A tiles (HBM)   "), (1, "shared A tile (HBM)     "), (2, "distinct A tiles (L2 set)"), (3, "shared A tile (L2 set)  ")):
        def f():
            census.zero_()
            rc = pl.probe_dma_share(A.data_ptr(), W.data_ptr(), a_tile, w_tile, ntile, ksteps, mode, census.data_ptr(), st)
            assert rc == 0
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        byts = 512 * ntile * ksteps * 24576
        print(f"   {name}: {us:8.1f} us  {byts / us / 1e6:6.2f} TB/s into LDS")
