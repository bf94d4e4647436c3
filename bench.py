#!/usr/bin/env python3
"""Training-throughput benchmark (images/sec, fwd + loss + bwd + all-reduce + AdamW) on N MI355X, one
process per GPU (RCCL over xGMI).  The default is the headline, BASELINE.json configs[1]:

    python bench.py --gpus 1 --steps 20 --warmup 5                      # ViT-MHLA-Small 224/p16, 256 img/GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

--config selects the other BASELINE.json configurations:
    cfg1  ViT-Tiny (models/vit.py, dense attention) 32x32 patch4, 64 img/GPU, step replayed from a HIP graph;
          CPU baseline at the exact B = 64
    cfg2  ViT-MHLA-Small 224x224 patch16, 197 tokens, 256 img/GPU            (headline, default)
    cfg3  SPPP+MHLA Small 224x224, 16 superpixels -> 17 tokens, 128 img/GPU, step replayed from a HIP graph
    cfg4  ViT-MHLA-Base 384x384 patch16, 577 tokens, 64 img/GPU, --dtype bf16 | fp8

A step = zero_grad -> forward -> cross-entropy -> backward -> gradient all-reduce -> AdamW on one synthetic
batch already resident in HBM.  Rank 0 prints ONE JSON line.  Per-launch GEMM timing (HIP events on the launch
stream) runs in extra, un-timed steps AFTER the timed region.
"""
import argparse
import glob
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0
BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
F32_MFMA_PEAK_TFLOPS = 157.3
FP8_DENSE_PEAK_TFLOPS = 5000.0
PROFILE_DIR = os.path.join(ROOT, "profiles")


def traffic_profile(cfg, dtype):
    """The newest committed PMC traffic profile of this configuration / dtype (profiles/rNN_pmc_hbm_traffic*.json,
    written by tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)."""
    best = None
    for f in sorted(glob.glob(os.path.join(PROFILE_DIR, "r*_pmc_hbm_traffic*.json"))):
        try:
            with open(f) as fh:
                prof = json.load(fh)
        except (OSError, ValueError):
            continue
        if prof.get("config", "cfg2") == cfg and prof.get("dtype", "bf16") == dtype:
            best = (f, prof)
    return best


def match_gemm_kernels(prof, key, tag):
    """Kernels of a traffic profile that belong to the GEMM family `key` (dtype_layout_out, kernels.GEMM_TRACE) as the
    library dispatched it (`tag`: favit_gemm_last_kernel).  Returns (bytes per launch averaged over launches, names)."""
    base = {"p4": "gemm_fp8_p4_kernel" if key.startswith("fp8") else "gemm_bf16_p4_kernel", "p7": "gemm_bf16_p7_kernel",
            "pp": "gemm_bf16_pp_kernel", "s64": "gemm_bf16_s64_kernel", "grouped_tn": "gemm_bf16_p4_grouped_tn_kernel",
            "t128": "gemm_"}.get(tag, "gemm_")
    ak, bk = key.split("_")[1][0] == "K", key.split("_")[1][1] == "K"
    of32 = key.endswith("of32") or key.endswith("grouped")
    tot_b = tot_n = 0.0
    names = []
    for e in prof.get("kernels", []):
        n = e["kernel"]
        if base not in n or ("grouped" in n) != (tag == "grouped_tn"):
            continue
        demangled = "<" in n
        if tag in ("p4", "t128") and not key.startswith("fp8"):
            flags = (f"<{str(ak).lower()}, {str(bk).lower()}," if demangled else f"ILb{int(ak)}ELb{int(bk)}E")
            if flags not in n:
                continue
        if tag in ("s64", "pp"):
            flag = (f"<{str(bk).lower()}," if demangled else f"ILb{int(bk)}E")
            if flag not in n:
                continue
        is_f32 = ("float" in n.split("(")[0]) if demangled else (n.rstrip("E").endswith("f") or "EfE" in n or "IfE" in n)
        if is_f32 != of32:
            continue
        tot_b += e["hbm_bytes_per_launch"] * e["launches_per_step"]
        tot_n += e["launches_per_step"]
        names.append(n[:60])
    return (tot_b / tot_n if tot_n else None), names


def flops_per_image_train(L, D, depth, hd, W, N, P, C, classes, dense=False):
    """BASELINE.md section 3 / SURVEY 8(d): F_train = 3 * F_fwd (1 MAC = 2 FLOP)."""
    blk = L * (24 * D * D + 4 * L * D) if dense else L * (24 * D * D + 4 * D * hd + 4 * W * D)
    fwd = depth * blk + 2 * N * P * P * C * D + 2 * D * classes
    return 3 * fwd


def synthetic_label_maps(n_maps, img, regions, seed):
    """Voronoi label maps (input data of the SPPP configurations: SLIC itself is outside the path).  Jittered
    grid seeds so that every region dominates at least one 16x16 patch."""
    rs = np.random.RandomState(seed)
    side = int(round(regions ** 0.5))
    yy, xx = np.mgrid[0:img, 0:img]
    out = []
    for _ in range(n_maps):
        cell = img / side
        cy, cx = np.mgrid[0:side, 0:side]
        py = (cy.ravel() + 0.5 + rs.uniform(-0.2, 0.2, side * side)) * cell
        px = (cx.ravel() + 0.5 + rs.uniform(-0.2, 0.2, side * side)) * cell
        d = (yy[None] - py[:, None, None]) ** 2 + (xx[None] - px[:, None, None]) ** 2
        out.append(d.argmin(0).astype(np.int64))
    return np.stack(out)


CONFIGS = {
    "cfg1": dict(name="ViT-Tiny (D192/12L/3H, dense attention) 32x32 patch4, 65 tokens", batch=64, img=32, classes=10,
                 flops=dict(L=65, D=192, depth=12, hd=64, W=0, N=64, P=4, C=3, classes=10, dense=True),
                 metric="images/sec (train fwd+bwd) ViT-Tiny 32/p4", cpu_batch=64, cpu_steps=24),
    "cfg2": dict(name="ViT-MHLA-Small (D384/12L/6H, window 7) 224x224 patch16, 197 tokens", batch=256, img=224,
                 classes=1000, flops=dict(L=197, D=384, depth=12, hd=64, W=7, N=196, P=16, C=3, classes=1000),
                 metric="images/sec (train fwd+bwd) ViT-MHLA 224/p16", cpu_batch=16, cpu_steps=24),
    "cfg3": dict(name="SPPP+MHLA Small (D384/12L/6H, 16 superpixels -> 17 tokens, window 7) 224x224 patch16, "
                      "HIP-graph replayed step", batch=128, img=224, classes=1000,
                 flops=dict(L=17, D=384, depth=12, hd=64, W=7, N=196, P=16, C=3, classes=1000),
                 metric="images/sec (train fwd+bwd) SPPP+MHLA 224/p16", cpu_batch=16, cpu_steps=12),
    "cfg4": dict(name="ViT-MHLA-Base (D768/12L/12H, window 7) 384x384 patch16, 577 tokens", batch=64, img=384,
                 classes=1000, flops=dict(L=577, D=768, depth=12, hd=64, W=7, N=576, P=16, C=3, classes=1000),
                 metric="images/sec (train fwd+bwd) ViT-MHLA-Base 384/p16", cpu_batch=2, cpu_steps=4),
    # experiments/sppp_mhla_pretrained.py:236-247,337-346: identity latent_proj, everything frozen except head and
    # latent_proj, per-name learning rates, batches bucketed by superpixel-token count (R = 16 and R = 15 alternate)
    "cfg5": dict(name="SPPP+MHLA Small fine-tune (identity latent_proj; only head + latent_proj trainable; alternating "
                      "R = 16 / R = 15 superpixel-token buckets) 224x224 patch16, HIP-graph replayed steps", batch=128,
                 img=224, classes=1000, flops=dict(L=17, D=384, depth=12, hd=64, W=7, N=196, P=16, C=3, classes=1000),
                 metric="images/sec (fine-tune fwd+bwd) SPPP+MHLA 224/p16", cpu_batch=16, cpu_steps=12),
}


def build_model(pkg, cfg, dev, dropout=0.0, attn_dropout=0.0, embed_dropout=0.0):
    """The reference's dropout defaults are dropout 0.1, attn_dropout 0.0, embed_dropout 0.0 (main.py:106-111,
    experiments/mhla_pretrained.py:49-50): --dropout sets only the first; the other two have flags of their own."""
    M = pkg.models
    d, da, de = float(dropout), float(attn_dropout), float(embed_dropout)
    if cfg == "cfg1":
        return M.vit.VisionTransformer(img_size=32, patch_size=4, num_classes=10, embed_dim=192, depth=12, num_heads=3).to(dev)
    if cfg == "cfg2":
        return M.vit_mhla.VisionTransformerMHLA(img_size=224, patch_size=16, num_classes=1000, embed_dim=384, depth=12,
                                                num_heads=6, window_size=7, use_mhla=True, dropout=d,
                                                attn_dropout=da, embed_dropout=de).to(dev)
    if cfg in ("cfg3", "cfg5"):
        m = M.sppp_mhla.SPPPViTMHLA(img_size=224, patch_size=16, num_classes=1000, embed_dim=384, depth=12, num_heads=6,
                                    num_superpixels=16, pooling_type="mean", window_size=7, use_mhla=True)
        if cfg == "cfg5":
            for blk in m.blocks:                       # experiments/sppp_mhla_pretrained.py:236-237
                torch.nn.init.eye_(blk.attn.latent_proj.weight)
                torch.nn.init.zeros_(blk.attn.latent_proj.bias)
            for name, p_ in m.named_parameters():      # :243-247 (freeze_layers=True)
                if not any(x in name for x in ("head", "latent_proj", "segmentation", "patch_mapper", "pooling")):
                    p_.requires_grad = False
        return m.to(dev)
    return M.vit_mhla.VisionTransformerMHLA(img_size=384, patch_size=16, num_classes=1000, embed_dim=768, depth=12,
                                            num_heads=12, window_size=7, use_mhla=True, dropout=d, attn_dropout=da,
                                            embed_dropout=de).to(dev)


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


def active_knobs():
    """FAVIT_* environment variables set for this process: kernel-selection / debugging switches the library or the
    package reads.  A clean measurement has none."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("FAVIT_")}


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def csrc_sha16():
    """Fingerprint of the kernel sources: a PMC profile is only valid for the kernels it was taken on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "focused-attention-vit_amd", "csrc", "*"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_baseline(cfg, model, segs_np):
    """The CPU oracle (oracle/favit_oracle.py, a port of the reference's PyTorch-CPU path) timed on this box's
    host cores: same model configuration and weights, fp32, 1 warm-up + N timed fwd+bwd steps (bounded sample)."""
    from oracle import favit_oracle as O
    c = CONFIGS[cfg]
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline on {cores} host cores")
    batch, steps = c["cpu_batch"], c["cpu_steps"]
    trainable = {k for k, p_ in model.named_parameters() if p_.requires_grad}
    sd = {k: v.detach().float().cpu().clone().requires_grad_(k in trainable) for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(batch, 3, c["img"], c["img"], generator=g)
    y = torch.randint(0, c["classes"], (batch,), generator=g)

    def fwd():
        if cfg == "cfg1":
            return O.vit_forward(x, sd, 4, 3)
        if cfg == "cfg2":
            return O.vit_mhla_forward(x, sd, 16, 6, 7, True)
        if cfg in ("cfg3", "cfg5"):
            return O.sppp_vit_mhla_forward(x, segs_np[:batch], sd, 16, 6, 7, True)
        return O.vit_mhla_forward(x, sd, 16, 12, 7, True)

    def step():
        for v in sd.values():
            v.grad = None
        O.cross_entropy(fwd(), y).backward()

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{c['name'].split(' (')[0]} fwd+bwd fp32, batch {batch}, 1 warm-up + {steps} timed steps "
                      f"({dt:.1f} s of CPU work)"}


def main():
    if os.environ.get("FAVIT_BENCH_WATCHDOG"):      # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        import threading
        limit = int(os.environ["FAVIT_BENCH_WATCHDOG"])
        faulthandler.dump_traceback_later(limit, exit=True)

        def native_threads():
            # faulthandler only sees Python threads; the collective backends' workers are native.  Shortly before the
            # limit, record every OS thread of this process: name, scheduler state and the kernel wait channel
            # (futex = parked on a condition / HIP signal, poll / recv = network, running = spinning).
            time.sleep(max(1, limit - 5))
            out = os.path.join(ROOT, "gpurun_out", f"hang_rank{os.environ.get('RANK', '0')}.txt")
            os.makedirs(os.path.dirname(out), exist_ok=True)
            with open(out, "w") as f:
                for tid in sorted(os.listdir("/proc/self/task"), key=int):
                    rec = [tid]
                    for name in ("comm", "wchan", "syscall"):
                        try:
                            with open(f"/proc/self/task/{tid}/{name}") as g:
                                rec.append(g.read().strip()[:60])
                        except OSError as e:
                            rec.append(f"<{e.errno}>")
                    try:
                        with open(f"/proc/self/task/{tid}/stat") as g:
                            rec.append(g.read().rsplit(")", 1)[1].split()[0])
                    except OSError:
                        rec.append("?")
                    f.write("\t".join(rec) + "\n")
        threading.Thread(target=native_threads, daemon=True).start()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS), help="BASELINE.json configuration (cfg2 = headline)")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default: the configuration's own)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="bf16 = the headline; fp8 = the BASELINE.json configs[3] GEMM path (cfg4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default=os.environ.get("FAVIT_DIST_BACKEND", "nccl"),
                    help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing ranks on one GPU)")
    ap.add_argument("--no-gemm-trace", action="store_true")
    ap.add_argument("--slic-noise", action="store_true", help="--slic on the step's N(0,1) noise images (worst case)")
    ap.add_argument("--graph", action="store_true", help="cfg2 / cfg4: replay the step from HIP graphs as well")
    ap.add_argument("--no-graph", action="store_true", help="cfg1 / cfg3: eager launches instead of the replayed HIP graph")
    ap.add_argument("--side-stream", action="store_true", help="run weight-gradient GEMMs on a second HIP stream")
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="`dropout` of the cfg2 / cfg4 models: MLP, projection and residual-branch sites (the reference's "
                         "main.py:106 trains with 0.1 and attn_dropout = embed_dropout = 0.0; the headline row is 0.0, SURVEY 8d)")
    ap.add_argument("--attn-dropout", type=float, default=0.0, help="`attn_dropout` (reference default 0.0, main.py:108)")
    ap.add_argument("--embed-dropout", type=float, default=0.0, help="`embed_dropout` (reference default 0.0, main.py:110)")
    ap.add_argument("--bucket-mb", type=float, default=0.0, help="all-reduce bucket size in MiB (default: dp.GradSync's own choice)")
    ap.add_argument("--slic-cus", type=int, default=None,
                    help="--slic: CUs the side stream of the overlapped segmentation may use (default: DeviceLoader's)")
    ap.add_argument("--slic", action="store_true",
                    help="cfg3: also time the step WITH the device SLIC inside it (label maps recomputed from the batch every "
                         "step instead of installed once); reported as `slic_inclusive`, never as `value`")
    ap.add_argument("--host-input", action="store_true",
                    help="also measure the PCIe-inclusive rate: uint8 HWC batches in pinned HOST memory -> async copy + "
                         "device transform (data.DeviceLoader) -> step; reported as `pcie_inclusive`, never as `value`")
    args = ap.parse_args()
    c = CONFIGS[args.config]

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this host driver (before any HIP call)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local % ndev)
    torch.cuda.set_device(dev)
    force_dp = bool(os.environ.get("FAVIT_DP_FORCE"))          # one rank, but every collective is issued (RCCL call path)
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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
    model = build_model(pkg, args.config, dev, args.dropout, args.attn_dropout, args.embed_dropout)
    model.train()
    health = pkg.train.Health(dev)                   # first non-finite loss / gradient / parameter, noted by the kernels
    B = args.batch or c["batch"]
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn(B, 3, c["img"], c["img"], device=dev, generator=g)
    labels = torch.randint(0, c["classes"], (B,), device=dev, generator=g)
    segs_np = None
    segs = segs15 = None
    if args.config in ("cfg3", "cfg5"):
        segs_np = synthetic_label_maps(8, 224, 16, seed=100 + rank)
        segs = torch.from_numpy(np.stack([segs_np[i % 8] for i in range(B)])).to(dev)
        model.segmentation.set_label_maps(segs)
        if args.config == "cfg5":                    # the R = 15 bucket: regions 14 and 15 merged
            segs15 = torch.where(segs == 15, torch.full_like(segs, 14), segs)
        segs_np = np.stack([segs_np[i % 8] for i in range(max(B, c["cpu_batch"]))])
    if args.config == "cfg5":
        # the reference's frozen-layers branch gives every trainable tensor its own group: head at head_lr, the rest
        # (latent_proj) at lr (experiments/sppp_mhla_pretrained.py:337-346)
        groups = pkg.train.param_groups(model, lr=1e-4, head_lr=1e-3, latent_lr_mult=1.0)
    else:
        groups = pkg.train.param_groups(model, lr=1e-4)
    opt = pkg.train.FusedAdamW(groups, lr=1e-4, weight_decay=0.05, bucket_mb=args.bucket_mb or None)
    graphed = (args.config in ("cfg1", "cfg3", "cfg5") or args.graph) and not args.no_graph
    gstep = None
    if args.config == "cfg5":
        buckets = []
        for maps, R in ((segs, 16), (segs15, 15)):
            model.segmentation.set_label_maps(maps)
            model.assume_num_tokens = R
            if graphed:
                g_ = pkg.train.GraphedStep(model, opt, images, labels, static_inputs=True)   # (the batch is resident in HBM)
                buckets.append(lambda g_=g_: g_(images, labels))
            else:
                def eager(maps=maps, R=R):
                    model.segmentation.set_label_maps(maps)
                    model.assume_num_tokens = R
                    return pkg.train.train_step(model, images, labels, opt)
                buckets.append(eager)
        turn = {"i": 0}

        def step():
            turn["i"] += 1
            return buckets[turn["i"] % 2]()
    elif graphed:
        # hundreds of launches of a few microseconds per step: the Python launch path, not the GPU, would set the
        # step time (cfg3: 17 tokens per image; cfg1: 65 tokens of width 192)
        if args.config == "cfg3":
            model.assume_num_tokens = 16
        gstep = pkg.train.GraphedStep(model, opt, images, labels, static_inputs=True)       # (the batch is resident in HBM)
        step = lambda: gstep(images, labels)
    else:
        step = lambda: pkg.train.train_step(model, images, labels, opt)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: {args.config} model built, warming up")
    for _ in range(args.warmup):
        step()
    sync()
    log("warm-up done, timing")

    e_begin, e_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e_begin.record()
    poison_at = int(os.environ.get("FAVIT_BENCH_TEST_POISON_STEP", "-1"))   # tests only: exercises the non-finite report
    for s in range(args.steps):
        if s == poison_at:
            opt.groups[0]["flat"].flat_p[7].fill_(float("nan"))
        loss = step()
    e_end.record()
    sync()
    dt = time.perf_counter() - t0
    log(f"timed region done: {1e3 * dt / args.steps:.2f} ms/step")

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    loss_val = float(loss.item())
    # A poisoned run is not a measurement: name what went non-finite and where, keep the evidence, print NO metric line.
    bad = health.poll()
    if bad is not None or not np.isfinite(loss_val):
        rep = health.report(opt, model, extra=[("loss", loss)]) or {}
        rep.update({"loss": loss_val, "config": args.config, "dtype": args.dtype, "steps": args.steps, "warmup": args.warmup,
                    "rank": rank, "world": world, "hip_graph": bool(graphed), "seed": 1234 + rank,
                    "adamw_launches_per_step": len(opt.groups), "knobs": active_knobs()})
        out_dir = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, f"nonfinite_{args.config}_rank{rank}_{int(time.time())}.json")
        with open(path, "w") as fh:
            json.dump(rep, fh, indent=1)
        log(f"NON-FINITE values in the step: {json.dumps(rep)}  (written to {path})")
        sys.exit(3)

    # PCIe-inclusive rate (informational): every step consumes a fresh uint8 batch from pinned host memory
    pcie = None
    if args.host_input and not graphed:
        D_ = pkg.data
        tf = D_.DeviceTransform("resize", c["img"], (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
        rs = np.random.RandomState(7 + rank)
        host = [(torch.from_numpy(rs.randint(0, 256, size=(B, c["img"], c["img"], 3), dtype=np.uint8)).pin_memory(),
                 torch.from_numpy(rs.randint(0, c["classes"], size=B).astype(np.int64)).pin_memory()) for _ in range(4)]
        n_it = max(4, args.steps)
        batches = [host[i % 4] for i in range(n_it + 2)]
        it = iter(D_.DeviceLoader(batches, tf, device=dev))
        for _ in range(2):
            xb, yb = next(it)
            pkg.train.train_step(model, xb, yb, opt)
        sync()
        tp0 = time.perf_counter()
        for xb, yb in it:
            pkg.train.train_step(model, xb, yb, opt)
        sync()
        tp = time.perf_counter() - tp0
        pcie = {"images_per_sec": round(world * B * n_it / tp, 2), "ms_per_step": round(1e3 * tp / n_it, 3),
                "input": f"uint8 HWC {c['img']}x{c['img']}x3 from pinned host memory, async copy stream + device transform"}

    # SPPP end to end (informational): the label maps are recomputed from the batch by the device SLIC in EVERY step
    # (features + k-means + connectivity, ~10 launches) and copied into the buffer the replayed graph reads
    slic_inc = None
    if args.slic and args.config == "cfg3":
        # images with spatial correlation for the segmentation (a 14 x 14 random colour field upsampled bicubically and
        # normalised like the loader's output): on N(0,1) pixel noise every image falls into ~2,000 connected
        # components, the worst case of the connectivity stage, which no photograph produces (--slic-noise times that)
        if args.slic_noise:
            simg = images
        else:
            low = torch.rand(B, 3, 14, 14, device=dev, generator=g)
            simg = ((torch.nn.functional.interpolate(low, size=(c["img"], c["img"]), mode="bicubic").clamp(0, 1) - 0.5) / 0.5).contiguous()
        K.slic(simg, n_segments=16, compactness=10.0)              # warm-up (lazy attributes, allocator)
        sync()
        ts0 = time.perf_counter()
        sgm = model.segmentation
        for _ in range(args.steps):
            sgm.update_label_maps(K.slic(simg, n_segments=16, compactness=10.0))    # (+ the patch mapping / centroids)
            step()
        sync()
        ts = time.perf_counter() - ts0
        slic_inc = {"images_per_sec": round(world * B * args.steps / ts, 2), "ms_per_step": round(1e3 * ts / args.steps, 3),
                    "images": "N(0,1) pixel noise" if args.slic_noise else "smooth random colour fields, mean/std-normalised",
                    "what": "device SLIC (compactness 10, 16 segments) of a batch + the step, every step; the "
                            "superpixel-token count of synthetic images is not 16 for every image: the replayed graph "
                            "was captured for 16 tokens and the timing, not the loss, is what this line reports"}
        # the same with the SLIC of batch i + 1 on a side stream under step i (what a prefetching loader does)
        slic_cus = args.slic_cus if args.slic_cus is not None else pkg.data.SEGMENTER_CUS
        side = pkg.streams.cu_masked_stream(slic_cus)
        # a CU-masked stream is a BLOCKING stream (the API has no flag): it synchronises with the null stream in both
        # directions, so the step of this leg runs on a stream of torch's pool instead of the default one
        main = torch.cuda.Stream()
        main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main):
            ready, used = torch.cuda.Event(), torch.cuda.Event()
            nxt = K.slic(simg, n_segments=16, compactness=10.0)
            nxd = sgm.derive_for(nxt)
            ready.record(main)
            step()                                            # first replay on this stream outside the timed region
            sync()
            ts0 = time.perf_counter()
            for _ in range(args.steps):
                main.wait_event(ready)
                sgm.update_label_maps(nxt, nxd)               # copies into the tensors the captured step reads
                nxt.record_stream(main)
                for tens in nxd.values():
                    for t_ in tens:
                        t_.record_stream(main)
                used.record(main)
                step()                                        # the step first: its launch must not queue behind the
                with torch.cuda.stream(side):                 # host side of the ~40 segmentation launches
                    side.wait_event(used)
                    nxt = K.slic(simg, n_segments=16, compactness=10.0)
                    nxd = sgm.derive_for(nxt)                 # patch mapping + centroids of the next batch
                    ready.record(side)
            sync()
            ts = time.perf_counter() - ts0
        torch.cuda.current_stream().wait_stream(main)
        slic_inc["overlapped"] = {"images_per_sec": round(world * B * args.steps / ts, 2),
                                  "ms_per_step": round(1e3 * ts / args.steps, 3),
                                  "side_stream_cus": slic_cus,
                                  "what": "SLIC of the next batch on a CU-masked side stream while the step of this one runs"}

    # per-launch GEMM timing: two extra EAGER steps after the timed region (events cannot be captured in a graph).
    # Every rank runs them (the all-reduce inside opt.step() is collective); only rank 0 records events.
    trace, traced_steps, opt_ms = None, 0, None
    if not args.no_gemm_trace:
        trace = [] if rank == 0 else None
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for s in range(2):
            K.GEMM_TRACE = trace
            opt.zero_grad()
            e[0].record()
            lo = pkg.train.cross_entropy(model(images), labels)
            lo.backward()
            K.GEMM_TRACE = None
            e[1].record()
            opt.step()
            e[2].record()
            traced_steps += 1
        torch.cuda.synchronize()
        opt_ms = e[1].elapsed_time(e[2])
    if world > 1:
        dist.barrier()

    if rank == 0:
        n_img = world * B * args.steps
        out = {
            "metric": c["metric"],
            "value": round(n_img / dt, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{c['name']}, {B} images/GPU; step = fwd + cross-entropy + bwd + grad all-reduce + AdamW",
                       "global_batch": world * B, "parallelism": f"dp{world}", "weights": "random-init (seed 1234)",
                       "baseline_config": args.config, "hip_graph": bool(graphed), "dropout": args.dropout,
                       "attn_dropout": args.attn_dropout, "embed_dropout": args.embed_dropout},
            "knobs": active_knobs(),
            "gpu_ms_per_step_events": round(e_begin.elapsed_time(e_end) / args.steps, 3),
            "loss": round(loss_val, 5),
            "model_tflops_per_s": round(n_img * flops_per_image_train(**c["flops"]) / dt / 1e12, 2),
        }
        if opt_ms is not None:
            out["optimizer_and_allreduce_wait_ms"] = round(opt_ms, 3)
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        if slic_inc is not None:
            out["slic_inclusive"] = slic_inc
        if trace:
            fam = {}
            for e0, e1, fl, key, shp, kern in trace:
                d = fam.setdefault(key, [0.0, 0.0, 0])
                d[0] += e0.elapsed_time(e1) * 1e-3
                d[1] += fl
                d[2] += 1
            dom = max(fam, key=lambda k: fam[k][0])
            sec, fl, n = fam[dom]
            peak = (BF16_DENSE_PEAK_TFLOPS if dom.startswith("bf16") else
                    FP8_DENSE_PEAK_TFLOPS if dom.startswith("fp8") else F32_MFMA_PEAK_TFLOPS)
            ach = fl / sec / 1e12
            # HBM bytes per launch come from committed PMC passes (rocprofv3 cannot run inside this process); they
            # are only reported while the kernel sources are the ones the profile was taken on.
            traffic, stale, tsrc, tprof = None, None, None, None
            found = traffic_profile(args.config, args.dtype)
            if found is not None:
                tsrc, tprof = found
                stale = tprof.get("csrc_sha16") != csrc_sha16()
                tags = {}
                for t_ in trace:
                    if t_[3] == dom:
                        tags[t_[5]] = tags.get(t_[5], 0) + 1
                tag = max(tags, key=tags.get)
                traffic, _names = match_gemm_kernels(tprof, dom, tag)
                if traffic is None and dom in tprof.get("families", {}):
                    traffic = tprof["families"][dom]["hbm_bytes_per_launch"]
                traffic = None if traffic is None else int(traffic)
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": traffic,
                               "kernel": f"gemm_{dom}", "launches": n, "avg_launch_us": round(1e6 * sec / n, 2),
                               "avg_flops_per_launch": round(fl / n, 1),
                               "measured": "HIP events around every launch of this kernel family in 2 un-timed steps after the timed region"}
            if traffic is not None:
                out["roofline"]["traffic_source"] = os.path.relpath(tsrc, ROOT)
                out["roofline"]["traffic_stale"] = bool(stale)      # true: csrc/ changed since the PMC passes were taken
                gbps = traffic / (sec / n) / 1e9
                out["roofline"]["hbm_view"] = {"achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                               "frac": round(gbps / HBM_PEAK_GBPS, 4)}
            tot = sum(v[0] for v in fam.values())
            out["gemm_families"] = {k: {"ms_per_step": round(1e3 * v[0] / traced_steps, 3),
                                        "tflops": round(v[1] / v[0] / 1e12, 1), "launches_per_step": v[2] // traced_steps}
                                    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])}
            out["gemm_share_of_step"] = round(tot / traced_steps / (dt / args.steps), 3)
            # What bounds the STEP: the HBM bytes all of its kernels move (PMC, same profile) against the 8 TB/s peak
            # over the measured step time -- the step is HBM-bound as much as MFMA-bound.
            if tprof is not None and tprof.get("hbm_bytes_per_step"):
                hb = float(tprof["hbm_bytes_per_step"])
                out["step_hbm"] = {"bytes_per_step": int(hb), "achieved": round(hb / (dt / args.steps) / 1e9, 1),
                                   "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                   "frac": round(hb / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS, 4),
                                   "floor_ms_at_peak": round(hb / (HBM_PEAK_GBPS * 1e9) * 1e3, 3),
                                   "source": os.path.relpath(tsrc, ROOT), "stale": bool(stale),
                                   "what": "sum over every kernel of the step of (2*FETCH_SIZE + WRITE_SIZE) x launches"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, model, segs_np)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
