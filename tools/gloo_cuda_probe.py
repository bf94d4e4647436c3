#!/usr/bin/env python3
"""Rehearsal aid: N ranks on ONE GPU, gloo backend, several asynchronous all-reduces of slices of one flat CUDA buffer
per iteration with compute in between (the shape of GradSync's traffic, none of its logic).  If this hangs, a hang of
`bench.py --backend gloo` at the same rank count is gloo's CUDA path, not the data-parallel code."""
import faulthandler, os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port, n, sizes):
    faulthandler.dump_traceback_later(80, exit=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    flat = torch.ones(sum(sizes), device="cuda")
    a = torch.randn(4096, 4096, device="cuda")
    for it in range(n):
        hs = []; o = 0
        for s in sizes:
            b = a @ a                                   # compute between launches
            hs.append(dist.all_reduce(flat[o:o + s], async_op=True)); o += s
        for h in hs: h.wait()
        flat.mul_(1.0 / world)
        torch.cuda.synchronize()
        if rank == 0: print("iter", it, flat[0].item(), flush=True)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    world = int(sys.argv[1]); sizes = [2_900_000] * 8
    mp.spawn(w, args=(world, 29611, 8, sizes), nprocs=world, join=True)
