#!/usr/bin/env python3
"""What do the matrix cores of THIS device deliver under load?  A bare v_mfma_f32_16x16x32_bf16 loop (register
operands, random data, 1 / 2 / 4 waves per SIMD) -> TFLOP/s and the shader clock the chip held
(tools/probe_kernels.hip: mfma_peak_kernel).  Context for roofline.frac, which is priced against the 2.5 PF spec."""
import ctypes, os
import torch
pl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe.so"))
pl.probe_mfma_peak.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda"
seed_rand = torch.randn(4096 * 8, device=dev).to(torch.bfloat16)
seed_zero = torch.zeros(4096 * 8, device=dev, dtype=torch.bfloat16)
sink = torch.zeros(4, device=dev)
st = torch.cuda.current_stream().cuda_stream
iters = 20000
for name, seed in (("random", seed_rand), ("zeros ", seed_zero)):
    for wg_per_cu in (1, 2, 4):                      # 256-thread blocks: one wave per SIMD each
        blocks = 256 * wg_per_cu
        out = torch.zeros(2 * blocks, dtype=torch.int64, device=dev)
        def f():
            assert pl.probe_mfma_peak(seed.data_ptr(), iters, blocks, out.data_ptr(), sink.data_ptr(), st) == 0
        for _ in range(20): f()                      # hold the load long enough for the clock to settle
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 20
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        flops = blocks * 4 * iters * 16 * (2 * 16 * 16 * 32)
        o = out.view(-1, 2).double()
        ghz = (o[:, 0] / o[:, 1]).median().item() * 0.1
        print(f"{name} {wg_per_cu} wave(s)/SIMD: {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  clock {ghz:5.2f} GHz", flush=True)

# the exact-fp32 instruction (v_mfma_f32_32x32x2_f32: 4,096 flops per 64 cycles and SIMD; 157.3 TFLOP/s at 2.4 GHz)
pl.probe_mfma_peak_f32.argtypes = pl.probe_mfma_peak.argtypes
seed32 = torch.randn(4096, device=dev)
iters32 = 4000
for wg_per_cu in (1, 2, 4):
    blocks = 256 * wg_per_cu
    out = torch.zeros(2 * blocks, dtype=torch.int64, device=dev)
    def f32():
        assert pl.probe_mfma_peak_f32(seed32.data_ptr(), iters32, blocks, out.data_ptr(), sink.data_ptr(), st) == 0
    for _ in range(20): f32()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f32()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    flops = blocks * 4 * iters32 * 16 * (2 * 32 * 32 * 2)
    o = out.view(-1, 2).double()
    ghz = (o[:, 0] / o[:, 1]).median().item() * 0.1
    print(f"fp32 32x32x2, {wg_per_cu} wave(s)/SIMD: {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  clock {ghz:5.2f} GHz", flush=True)

# v_mfma_f32_16x16x4_f32: 2,048 flops per 32 cycles -- the same rate in shorter instructions
pl.probe_mfma_peak_f32s.argtypes = pl.probe_mfma_peak.argtypes
for wg_per_cu in (1, 2, 4):
    blocks = 256 * wg_per_cu
    out = torch.zeros(2 * blocks, dtype=torch.int64, device=dev)
    def f32s():
        assert pl.probe_mfma_peak_f32s(seed32.data_ptr(), iters32, blocks, out.data_ptr(), sink.data_ptr(), st) == 0
    for _ in range(20): f32s()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f32s()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    flops = blocks * 4 * iters32 * 16 * (2 * 16 * 16 * 4)
    o = out.view(-1, 2).double()
    ghz = (o[:, 0] / o[:, 1]).median().item() * 0.1
    print(f"fp32 16x16x4, {wg_per_cu} wave(s)/SIMD: {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  clock {ghz:5.2f} GHz", flush=True)
