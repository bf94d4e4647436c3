#!/usr/bin/env python3
"""Headline benchmark: training images/sec of ViT-MHLA-Small 224x224 / patch16 (197 tokens,
window 7) -- BASELINE.json configs[1] -- on N MI355X, one process per GPU (RCCL over xGMI).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = zero_grad -> forward -> cross-entropy -> backward -> gradient all-reduce -> AdamW on one
synthetic batch (images already resident in HBM).  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0
BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
F32_MFMA_PEAK_TFLOPS = 157.3
FP8_DENSE_PEAK_TFLOPS = 5000.0


def flops_per_image_train(L=197, D=384, depth=12, hd=64, W=7, N=196, P=16, C=3, classes=1000):
    """BASELINE.md section 3 / SURVEY 8(d): F_train = 3 * F_fwd."""
    blk = L * (24 * D * D + 4 * D * hd + 4 * W * D)
    fwd = depth * blk + 2 * N * P * P * C * D + 2 * D * classes
    return 3 * fwd


def host_cores():
    """CPU cores this process may actually use (affinity mask and cgroup quota, not the host total)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(pkg, model, batch=16, steps=24):
    """The CPU oracle (oracle/favit_oracle.py, a port of the reference's PyTorch-CPU path) timed on
    this box's host cores: same model config, B=16, 1 warm-up + `steps` timed fwd+bwd steps."""
    from oracle import favit_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline on {cores} host cores")
    sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(batch, 3, 224, 224, generator=g)
    y = torch.randint(0, 1000, (batch,), generator=g)

    def step():
        for v in sd.values():
            v.grad = None
        loss = O.cross_entropy(O.vit_mhla_forward(x, sd, 16, 6, 7, True), y)
        loss.backward()

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"ViT-MHLA-Small 224/p16 fwd+bwd fp32, batch {batch}, 1 warm-up + {steps} timed steps "
                      f"({dt:.1f} s of CPU work)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (BASELINE.json configs[1]: 256)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="bf16 = BASELINE.json configs[1] (the headline); fp8 = the configs[3] GEMM path, informational")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default=os.environ.get("FAVIT_DIST_BACKEND", "nccl"),
                    help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing ranks on one GPU)")
    ap.add_argument("--no-gemm-trace", action="store_true")
    ap.add_argument("--side-stream", action="store_true", help="run weight-gradient GEMMs on a second HIP stream")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this host driver (before any HIP call)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local % ndev)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    pkg = importlib.import_module("focused-attention-vit_amd")
    pkg._abi.lib()                                   # no HIP library -> fail loudly
    K = pkg.kernels
    pkg.set_compute_dtype(args.dtype)
    pkg.set_side_stream(args.side_stream)

    torch.manual_seed(1234)
    model = pkg.models.vit_mhla.VisionTransformerMHLA(img_size=224, patch_size=16, num_classes=1000, embed_dim=384,
                                                      depth=12, num_heads=6, window_size=7, use_mhla=True,
                                                      dropout=0.0, attn_dropout=0.0, embed_dropout=0.0).to(dev)
    model.train()
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn(B, 3, 224, 224, device=dev, generator=g)
    labels = torch.randint(0, 1000, (B,), device=dev, generator=g)
    opt = pkg.train.FusedAdamW(pkg.train.param_groups(model, lr=1e-4), lr=1e-4, weight_decay=0.05)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: model built, warming up")
    for _ in range(args.warmup):
        pkg.train.train_step(model, images, labels, opt)
    sync()
    log("warm-up done, timing")

    trace = None if args.no_gemm_trace else []
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
           torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    traced_steps = 0
    for s in range(args.steps):
        # HIP events around every favit_gemm launch cost ~5 % of a step: trace every 10th timed step
        K.GEMM_TRACE = trace if (trace is not None and s % 10 == 0) else None
        traced_steps += int(K.GEMM_TRACE is not None)
        ev[s][0].record()
        opt.zero_grad()
        loss = pkg.train.cross_entropy(model(images), labels)
        loss.backward()
        ev[s][1].record()
        opt.step()
        ev[s][2].record()
    sync()
    dt = time.perf_counter() - t0
    K.GEMM_TRACE = None
    log(f"timed region done: {1e3 * dt / args.steps:.2f} ms/step")

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss_val = float(loss.item())

    if rank == 0:
        n_img = world * B * args.steps
        fb = sum(a.elapsed_time(b) for a, b, _ in ev) / args.steps
        op_ms = sum(b.elapsed_time(c) for _, b, c in ev) / args.steps
        out = {
            "metric": "images/sec (train fwd+bwd) ViT-MHLA 224/p16",
            "value": round(n_img / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "ViT-MHLA-Small (D384/12L/6H, window 7) 224x224 patch16, 197 tokens, "
                                   f"{B} images/GPU; step = fwd + cross-entropy + bwd + grad all-reduce + AdamW",
                       "global_batch": world * B, "parallelism": f"dp{world}", "weights": "random-init (seed 1234)"},
            "breakdown_ms": {"fwd_bwd": round(fb, 3), "optimizer_and_allreduce_wait": round(op_ms, 3)},
            "loss": round(loss_val, 5),
            "model_tflops_per_s": round(n_img * flops_per_image_train() / dt / 1e12, 2),
        }
        if trace:
            fam = {}
            for e0, e1, fl, key, shp in trace:
                d = fam.setdefault(key, [0.0, 0.0, 0])
                d[0] += e0.elapsed_time(e1) * 1e-3
                d[1] += fl
                d[2] += 1
            dom = max(fam, key=lambda k: fam[k][0])
            sec, fl, n = fam[dom]
            peak = (BF16_DENSE_PEAK_TFLOPS if dom.startswith("bf16") else
                    FP8_DENSE_PEAK_TFLOPS if dom.startswith("fp8") else F32_MFMA_PEAK_TFLOPS)
            ach = fl / sec / 1e12
            traffic = None           # HBM bytes per launch from the committed PMC passes (profiles/)
            try:
                with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")) as f:
                    traffic = json.load(f)["families"][dom]["hbm_bytes_per_launch"]
            except (OSError, KeyError, ValueError):
                pass
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": traffic,
                               "kernel": f"gemm_{dom}", "launches": n, "avg_launch_us": round(1e6 * sec / n, 2),
                               "avg_flops_per_launch": round(fl / n, 1)}
            if traffic:
                # the same launches seen from the memory side (these K<=1536 GEMMs sit near the
                # 312 flop/B machine balance): measured HBM bytes / launch time vs the 8 TB/s peak
                gbps = traffic / (sec / n) / 1e9
                out["roofline"]["hbm_view"] = {"achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                               "frac": round(gbps / HBM_PEAK_GBPS, 4)}
            tot = sum(v[0] for v in fam.values())
            out["gemm_families"] = {k: {"ms_per_step": round(1e3 * v[0] / traced_steps, 3),
                                        "tflops": round(v[1] / v[0] / 1e12, 1), "launches_per_step": v[2] // traced_steps}
                                    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])}
            out["gemm_share_of_step"] = round(tot / traced_steps / (dt / args.steps), 3)
            out["gemm_traced_steps"] = traced_steps
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, model)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
