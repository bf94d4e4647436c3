#!/usr/bin/env python3
"""main.py-equivalent runner (the reference's main.py:64-149 cannot even be imported: three wrong
import names, main.py:41-43).  Same argument names for the settings that concern the path; the dataset is either a
.npz file {x_train uint8 [N,H,W,3], y_train, x_test, y_test} or a synthetic one (no downloads: there is no network).

    python tools/run_experiment.py --experiment mhla --img_size 32 --patch_size 4 --embed_dim 64 --depth 2 \
        --num_heads 4 --epochs 3 --batch_size 64 --results_dir gpurun_out/exp
"""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def synthetic_dataset(n, classes, size, seed):
    rs = np.random.RandomState(seed)
    protos = rs.randint(0, 256, size=(classes, size, size, 3))
    y = rs.randint(0, classes, size=n)
    x = np.clip(protos[y] + rs.randint(-40, 41, size=(n, size, size, 3)), 0, 255).astype(np.uint8)
    return x, y


def synthetic_blocks_dataset(n, classes, size, grid, seed, sample_seed=None):
    """Synthetic images for the SPPP experiment: a grid x grid board of flat colour blocks (a class-specific palette,
    per-sample brightness jitter per block).  SLIC finds exactly grid^2 superpixels on such an image, which is what the
    reference's forward needs from every image of a batch (torch.stack, models/sppp_mhla.py:300; the positional
    encoding accepts num_superpixels or num_superpixels - 1 regions, models/sppp.py:299) -- random-noise prototypes give
    1..16 regions per image and the reference's forward (and this one) raises on the first batch."""
    rs = np.random.RandomState(seed)
    # block colours well apart from their neighbours: a coarse lattice of base colours shuffled per class
    base = np.array([[r, g, b] for r in (40, 128, 215) for g in (40, 128, 215) for b in (40, 128, 215)], dtype=np.int32)
    pal = np.stack([base[rs.permutation(len(base))[:grid * grid]] for _ in range(classes)])        # [classes, g*g, 3]
    if sample_seed is not None:                          # (same palettes, other samples: the test split)
        rs = np.random.RandomState(sample_seed)
    y = rs.randint(0, classes, size=n)
    jit = rs.randint(-12, 13, size=(n, grid * grid, 1))
    blocks = np.clip(pal[y] + jit, 0, 255).astype(np.uint8).reshape(n, grid, grid, 3)
    cell = size // grid
    x = np.repeat(np.repeat(blocks, cell, axis=1), cell, axis=2)
    if x.shape[1] != size:                               # size not a multiple of the grid: pad by edge replication
        pad = size - x.shape[1]
        x = np.pad(x, ((0, 0), (0, pad), (0, pad), (0, 0)), mode="edge")
    return x, y


def batches(x, y, bs, shuffle, rs):
    idx = rs.permutation(len(x)) if shuffle else np.arange(len(x))
    return [(x[idx[i:i + bs]], y[idx[i:i + bs]]) for i in range(0, len(x) - bs + 1, bs)]


def main():
    ap = argparse.ArgumentParser(description="Vision Transformer experiments on the MI355X hot path")
    ap.add_argument("--experiment", required=True, choices=["traditional", "mhla", "sppp_mhla"])
    ap.add_argument("--data", default=None, help=".npz dataset (default: synthetic)")
    ap.add_argument("--dataset", default="cifar10", choices=["cifar10", "imagenet", "default"], help="transform stack")
    ap.add_argument("--results_dir", default="./results")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--img_size", type=int, default=224)
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--patch_size", type=int, default=16)
    ap.add_argument("--embed_dim", type=int, default=768)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--num_heads", type=int, default=12)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--num_superpixels", type=int, default=16)
    ap.add_argument("--compactness", type=float, default=0.1)
    ap.add_argument("--pooling_type", default="mean", choices=["mean", "max", "attention"])
    ap.add_argument("--window_size", type=int, default=7)
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--learning_rate", type=float, default=1e-4)
    ap.add_argument("--weight_decay", type=float, default=0.05)
    ap.add_argument("--head_learning_rate", type=float, default=1e-3)
    ap.add_argument("--compute_dtype", default="bf16", choices=["bf16", "fp32", "fp8"])
    ap.add_argument("--bucket_tokens", action="store_true",
                    help="sppp_mhla: run batches that mix images with num_superpixels and num_superpixels - 1 tokens group by "
                         "group (models.sppp.TokenBucketed; the reference -- and this tool without the flag -- fails on "
                         "such a batch in torch.stack, models/sppp_mhla.py:300; any other count fails in the "
                         "positional encoding, there and here)")
    a = ap.parse_args()

    pkg = importlib.import_module("focused-attention-vit_amd")
    pkg.set_compute_dtype(a.compute_dtype)
    torch.manual_seed(a.seed)
    rs = np.random.RandomState(a.seed)
    if a.data:
        d = np.load(a.data, allow_pickle=False)
        xtr, ytr, xte, yte = d["x_train"], d["y_train"], d["x_test"], d["y_test"]
    else:
        src = 32 if a.dataset == "cifar10" else a.img_size
        grid = int(round(a.num_superpixels ** 0.5))
        if a.experiment == "sppp_mhla" and grid * grid == a.num_superpixels:
            xtr, ytr = synthetic_blocks_dataset(2048, 10, src, grid, a.seed)
            xte, yte = synthetic_blocks_dataset(512, 10, src, grid, a.seed, sample_seed=a.seed + 1)
            if a.compactness == 0.1:
                # with the reference's colour-dominated default, blocks of similar colour merge and an image yields
                # 10..17 regions (measured); a spatially dominated SLIC returns the board's grid^2 cells for every image
                a.compactness = 200.0
                print("synthetic block images: SLIC compactness set to 200 (every image then yields "
                      f"{a.num_superpixels} superpixels; pass --compactness to override)")
        else:
            xtr, ytr = synthetic_dataset(2048, 10, src, a.seed)
            xte, yte = synthetic_dataset(512, 10, src, a.seed)       # same prototypes (same seed), fresh noise below
            xte = np.clip(xte.astype(np.int32) + rs.randint(-10, 11, size=xte.shape), 0, 255).astype(np.uint8)
    classes = int(max(ytr.max(), yte.max())) + 1
    M = pkg.models
    kw = dict(img_size=a.img_size, patch_size=a.patch_size, num_classes=classes, embed_dim=a.embed_dim, depth=a.depth,
              num_heads=a.num_heads, dropout=a.dropout)
    if a.experiment == "traditional":
        model = M.vit.VisionTransformer(**kw)
    elif a.experiment == "mhla":
        model = M.vit_mhla.VisionTransformerMHLA(window_size=a.window_size, use_mhla=True, **kw)
    else:
        model = M.sppp_mhla.SPPPViTMHLA(num_superpixels=a.num_superpixels, pooling_type=a.pooling_type,
                                        window_size=a.window_size, use_mhla=True, **kw)
        model.segmentation.compactness = a.compactness
    model = model.cuda()
    tfs = pkg.data.get_transforms(a.dataset, a.img_size, seed=a.seed)
    opt = pkg.train.FusedAdamW(pkg.train.param_groups(model, lr=a.learning_rate, head_lr=a.head_learning_rate),
                               lr=a.learning_rate, weight_decay=a.weight_decay, distributed=False)

    class Epochs:            # a fresh shuffle per epoch
        def __init__(self, x, y, shuffle):
            self.x, self.y, self.shuffle = x, y, shuffle
        def __len__(self):
            return len(self.x) // a.batch_size
        def __iter__(self):
            return iter(batches(self.x, self.y, a.batch_size, self.shuffle, rs))
    # SPPP: the loaders segment batch k + 1 (device SLIC on a CU-masked stream) under the step of batch k; the steps
    # then run on the loader's compute stream, because a CU-masked stream synchronises with the default stream
    seg = getattr(model, "segmentation", None)
    train_loader = pkg.data.DeviceLoader(Epochs(xtr, ytr, True), tfs["train"], segmenter=seg)
    test_loader = pkg.data.DeviceLoader(Epochs(xte, yte, False), tfs["test"], segmenter=seg)
    work = train_loader.compute_stream if seg is not None else torch.cuda.current_stream()
    work.wait_stream(torch.cuda.current_stream())
    run = pkg.models.sppp.TokenBucketed(model) if (a.bucket_tokens and seg is not None) else model
    with torch.cuda.stream(work):
        res = pkg.harness.fit(run, train_loader, test_loader, opt, a.epochs)
        ev = pkg.harness.evaluate(run, test_loader, a.batch_size)
    torch.cuda.current_stream().wait_stream(work)
    row = {"model": a.experiment, "img_size": a.img_size, "patch_size": a.patch_size, "embed_dim": a.embed_dim, "depth": a.depth,
           "num_heads": a.num_heads, "window_size": a.window_size, "total_parameters": sum(p.numel() for p in model.parameters()),
           "avg_epoch_time": res["avg_epoch_time"], "total_training_time": res["total_training_time"],
           "final_val_acc": res["final_val_acc"], "final_val_loss": res["final_val_loss"], "test_acc": ev["test_acc"],
           "test_loss": ev["test_loss"], "avg_inference_time_per_image": ev["avg_inference_time_per_image"],
           "peak_gpu_memory_mb": res["peak_gpu_memory_mb"]}
    path = os.path.join(a.results_dir, f"exp_{a.experiment}.csv")
    pkg.harness.save_results_csv(path, row)
    print(f"Results saved to {path}")


if __name__ == "__main__":
    main()
