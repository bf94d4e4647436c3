"""Measurement / experiment harness: counterparts of the reference's
``utils/metrics.py:152-308`` (measure_inference_time / measure_training_time / measure_memory_usage) and of the
epoch loop + results CSV of ``experiments/mhla_pretrained.py:330-525``.

Differences that matter on a GPU: the reference reads ``time.time()`` around asynchronous launches without ever
synchronising (utils/metrics.py:179-183), so its numbers are launch times; here every timed region is bracketed by
HIP events on the compute stream and one synchronisation.  The epoch loop accumulates loss / accuracy ON THE
DEVICE (the reference calls ``loss.item()`` and ``.sum().item()`` per batch: two host syncs per step)."""
from __future__ import annotations

import csv
import os
import time
from typing import Callable, Dict, Iterable, List, Optional

import torch

from . import train as T


def _sync_time(fn: Callable[[], None], iters: int) -> float:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3


def measure_inference_time(model, input_tensor, num_iterations: int = 100, warm_up: int = 250) -> Dict[str, float]:
    """utils/metrics.py:152-193 (same keys), timed with device events."""
    dev = next(model.parameters()).device
    x = input_tensor.to(dev)
    with torch.no_grad():
        for _ in range(warm_up):
            model(x)
        total = _sync_time(lambda: model(x), num_iterations)
    return {"total_time": total, "avg_time": total / num_iterations, "fps": num_iterations / total}


def measure_training_time(model, input_tensor, target, criterion, optimizer, num_iterations: int = 10) -> Dict[str, float]:
    """utils/metrics.py:196-240 (same keys).  optimizer: torch.optim.* or train.FusedAdamW."""
    dev = next(model.parameters()).device
    x, y = input_tensor.to(dev), target.to(dev)

    def step():
        optimizer.zero_grad()
        loss = criterion(model(x), y)
        loss.backward()
        optimizer.step()
    total = _sync_time(step, num_iterations)
    return {"total_time": total, "avg_time": total / num_iterations, "iterations_per_second": num_iterations / total}


def measure_memory_usage(model, input_tensor, backward: bool = False) -> Dict[str, float]:
    """utils/metrics.py:243-308 (same keys; psutil for the host side)."""
    import psutil
    dev = next(model.parameters()).device
    x = input_tensor.to(dev)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    proc = psutil.Process(os.getpid())
    cpu0, gpu0 = proc.memory_info().rss, torch.cuda.memory_allocated()
    if backward:
        model(x).sum().backward()
    else:
        with torch.no_grad():
            model(x)
    torch.cuda.synchronize()
    cpu1, gpu1, peak = proc.memory_info().rss, torch.cuda.memory_allocated(), torch.cuda.max_memory_allocated()
    mb = 1024 * 1024
    return {"cpu_memory_before_bytes": cpu0, "cpu_memory_after_bytes": cpu1, "cpu_memory_used_bytes": cpu1 - cpu0,
            "cpu_memory_used_mb": (cpu1 - cpu0) / mb, "gpu_memory_before_bytes": gpu0, "gpu_memory_after_bytes": gpu1,
            "gpu_memory_used_bytes": gpu1 - gpu0, "gpu_memory_used_mb": (gpu1 - gpu0) / mb,
            "gpu_memory_peak_bytes": peak, "gpu_memory_peak_mb": peak / mb}


def evaluate(model, loader: Iterable, batch_size: Optional[int] = None) -> Dict[str, float]:
    """experiments/mhla_pretrained.py:436-484: test loss / accuracy / inference time per image."""
    model.eval()
    dev = next(model.parameters()).device
    loss_sum = torch.zeros((), device=dev)
    correct = torch.zeros((), device=dev, dtype=torch.int64)
    total = n_batches = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    infer_s = 0.0
    with torch.no_grad():
        for images, labels in loader:
            e0.record()
            out = model(images)
            e1.record()
            loss_sum += T.cross_entropy(out, labels)
            correct += (out.argmax(1) == labels).sum()
            total += labels.numel()
            n_batches += 1
            e1.synchronize()
            infer_s += e0.elapsed_time(e1) * 1e-3
    n_batches = max(1, n_batches)
    bs = batch_size or max(1, total // n_batches)
    return {"test_loss": loss_sum.item() / n_batches, "test_acc": 100.0 * correct.item() / max(1, total),
            "avg_inference_time": infer_s / n_batches, "avg_inference_time_per_image": infer_s / n_batches / bs}


def fit(model, train_loader: Iterable, val_loader: Optional[Iterable], optimizer, epochs: int,
        log: Callable[[str], None] = print) -> Dict[str, object]:
    """The reference's epoch loop (experiments/mhla_pretrained.py:350-420) with device-side statistics: one host
    sync per EPOCH.  optimizer: train.FusedAdamW (hot path) or any torch.optim optimizer."""
    dev = next(model.parameters()).device
    hist: Dict[str, List[float]] = {"train_loss": [], "train_acc": [], "val_loss": [], "val_acc": [], "epoch_time": []}
    t_start = time.perf_counter()
    peak_mb = 0.0
    for ep in range(epochs):
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        model.train()
        loss_sum = torch.zeros((), device=dev)
        correct = torch.zeros((), device=dev, dtype=torch.int64)
        total = n_batches = 0
        for images, labels in train_loader:
            optimizer.zero_grad()
            out = model(images)
            loss = T.cross_entropy(out, labels)
            loss.backward()
            optimizer.step()
            loss_sum += loss.detach()
            correct += (out.detach().argmax(1) == labels).sum()
            total += labels.numel()
            n_batches += 1
        hist["train_loss"].append(loss_sum.item() / max(1, n_batches))          # the epoch's single sync
        hist["train_acc"].append(100.0 * correct.item() / max(1, total))
        if val_loader is not None:
            ev = evaluate(model, val_loader)
            hist["val_loss"].append(ev["test_loss"])
            hist["val_acc"].append(ev["test_acc"])
        torch.cuda.synchronize()
        hist["epoch_time"].append(time.perf_counter() - t0)
        peak_mb = max(peak_mb, torch.cuda.max_memory_allocated() / (1024 * 1024))
        log(f"Epoch {ep + 1}/{epochs} | Train Loss: {hist['train_loss'][-1]:.4f} | Train Acc: {hist['train_acc'][-1]:.2f}% | "
            + (f"Val Loss: {hist['val_loss'][-1]:.4f} | Val Acc: {hist['val_acc'][-1]:.2f}% | " if val_loader is not None else "")
            + f"Time: {hist['epoch_time'][-1]:.2f}s")
    return {"history": hist, "avg_epoch_time": sum(hist["epoch_time"]) / max(1, epochs),
            "total_training_time": time.perf_counter() - t_start,
            "final_val_acc": hist["val_acc"][-1] if hist["val_acc"] else float("nan"),
            "final_val_loss": hist["val_loss"][-1] if hist["val_loss"] else float("nan"), "peak_gpu_memory_mb": peak_mb}


def save_results_csv(path: str, row: Dict[str, object]) -> None:
    """One-row results CSV like experiments/mhla_pretrained.py:486-525 (pandas-free)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(row.keys()))
        w.writeheader()
        w.writerow(row)
