"""The overlapped data-parallel path through the REAL kernels: two ranks (gloo backend, both on the
one GPU of the test box -- RCCL needs one GPU per rank) run forward / backward of ViT-MHLA on half a batch each with
direct gradient accumulation into the flat buffers, kernel-side grad_ready calls, bucket all-reduces launched during
backward and FusedAdamW's finish(average=False) / grad_scale = 1/world.  The averaged gradients must equal the
single-process gradients of the whole batch, and after two optimizer steps both ranks must hold the same weights.
(autograd also fires the post-accumulate hook of a parameter whose gradient the kernels wrote directly; GradSync must
not count that as a second contribution -- the regression this test pins.)"""
import importlib
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _model(pkg):
    torch.manual_seed(3)
    return pkg.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2,
                                                     num_heads=4, window_size=7, use_mhla=True).cuda().train()


def _batch():
    g = torch.Generator().manual_seed(5)
    return torch.randn(8, 3, 32, 32, generator=g), torch.randint(0, 10, (8,), generator=g)


def _worker(rank, world, port, out, mode):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["FAVIT_DP_VERIFY"] = "1"                     # ordering check of every bucket against its launch-time snapshot
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("focused-attention-vit_amd")
    pkg.set_compute_dtype(mode)
    m = _model(pkg)
    opt = pkg.train.FusedAdamW(pkg.train.param_groups(m, lr=1e-2), lr=1e-2, weight_decay=0.0, bucket_mb=0.05)
    assert all(g["sync"] is not None and len(g["sync"].buckets) >= 1 for g in opt.groups)
    x, y = _batch()
    lo = rank * 4
    xs, ys = x[lo:lo + 4].cuda(), y[lo:lo + 4].cuda()
    opt.zero_grad()
    pkg.train.cross_entropy(m(xs), ys).backward()
    launched = [all(g["sync"]._launched) for g in opt.groups]           # every bucket went out DURING backward
    for g in opt.groups:
        g["sync"].finish(average=False)
    grads = {k: (p.grad / world).detach().cpu().clone() for k, p in m.named_parameters()}
    # two real optimizer steps: the ranks must stay in lock-step
    for _ in range(2):
        pkg.train.train_step(m, xs, ys, opt)
    torch.cuda.synchronize()
    w = torch.cat([p.detach().flatten().cpu() for p in m.parameters()])
    torch.save({"grads": grads, "w": w, "launched": launched}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
def test_dp_two_ranks_through_the_kernels(favit, tmp_path, mode, tol):
    out = str(tmp_path / "dp")
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out, mode), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0", weights_only=True), torch.load(out + ".1", weights_only=True)
    assert all(r0["launched"]) and all(r1["launched"])
    assert torch.equal(r0["w"], r1["w"]), "the two ranks diverged"
    favit.set_compute_dtype(mode)
    try:
        m = _model(favit)
        x, y = _batch()
        favit.train.cross_entropy(m(x.cuda()), y.cuda()).backward()
        for k, p in m.named_parameters():
            ref = p.grad.detach().cpu()
            for r in (r0, r1):
                err = (r["grads"][k] - ref).norm() / max(ref.norm().item(), 1e-12)
                assert err < tol, (k, float(err))
    finally:
        favit.set_compute_dtype("fp32")


def _nccl_worker(rank, port, out, wire=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["FAVIT_DP_FORCE"] = "1"                      # one rank, but issue every collective
    os.environ["FAVIT_DP_VERIFY"] = "1"                     # and check each against its launch-time snapshot (dp.py)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pkg = importlib.import_module("focused-attention-vit_amd")
    pkg.set_compute_dtype("bf16")
    m = _model(pkg)
    opt = pkg.train.FusedAdamW(pkg.train.param_groups(m, lr=1e-2), lr=1e-2, weight_decay=0.0, bucket_mb=0.05,
                               wire_dtype=wire)
    n_buckets = sum(len(g["sync"].buckets) for g in opt.groups)
    x, y = _batch()
    xs, ys = x.cuda(), y.cuda()
    opt.zero_grad()
    pkg.train.cross_entropy(m(xs), ys).backward()
    launched = [all(g["sync"]._launched) for g in opt.groups]
    n_handles = sum(len(g["sync"]._handles) for g in opt.groups)
    for g in opt.groups:
        g["sync"].finish(average=False)
    grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
    losses = [float(pkg.train.train_step(m, xs, ys, opt)) for _ in range(3)]
    torch.cuda.synchronize()
    verified = sum(getattr(g["sync"], "verified", 0) for g in opt.groups)
    torch.save({"grads": grads, "launched": launched, "n_handles": n_handles, "n_buckets": n_buckets, "losses": losses,
                "verified": verified}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_call_path_single_rank(favit, tmp_path):
    """RCCL itself (backend "nccl") on the one GPU of the box: a one-rank group in which GradSync still issues every
    bucket's asynchronous all-reduce from the autograd thread during backward (FAVIT_DP_FORCE), waits on the handles
    and feeds the fused AdamW.  A one-rank sum is the identity, so gradients must equal the plain single-process ones;
    what this covers is everything around the collective that a multi-GPU run uses (library load, communicator
    creation with device_id, stream ordering of the flat-buffer slices, handle waits, barrier)."""
    out = str(tmp_path / "nccl.pt")
    mp.spawn(_nccl_worker, args=(34500 + (os.getpid() % 2000), out), nprocs=1, join=True)
    r = torch.load(out, weights_only=True)
    assert all(r["launched"]) and r["n_handles"] == r["n_buckets"] >= 3
    assert r["verified"] >= 4 * r["n_buckets"], "FAVIT_DP_VERIFY: every bucket of every step equals its ordered snapshot"
    assert r["losses"][-1] < r["losses"][0]
    favit.set_compute_dtype("bf16")
    try:
        m = _model(favit)
        x, y = _batch()
        favit.train.cross_entropy(m(x.cuda()), y.cuda()).backward()
        for k, p in m.named_parameters():
            ref = p.grad.detach().cpu()
            err = (r["grads"][k] - ref).norm() / max(ref.norm().item(), 1e-12)
            assert err < 2e-2, (k, float(err))
    finally:
        favit.set_compute_dtype("fp32")


def test_rccl_bf16_wire_mode_single_rank(favit, tmp_path):
    """The opt-in bf16 exchange (dp.GradSync wire_dtype) on RCCL: bf16 all-reduce of the wire buffer's slices on the
    one-rank group, widened back into the fp32 gradient buffer.  A one-rank sum is the identity: the gradients the
    optimizer sees are the single-process ones rounded to bf16 once (and training still converges)."""
    out = str(tmp_path / "nccl_bf16.pt")
    mp.spawn(_nccl_worker, args=(36500 + (os.getpid() % 2000), out, torch.bfloat16), nprocs=1, join=True)
    r = torch.load(out, weights_only=True)
    assert all(r["launched"]) and r["n_handles"] == r["n_buckets"] >= 3
    assert r["losses"][-1] < r["losses"][0]
    rounded = 0
    for k, g in r["grads"].items():
        assert torch.isfinite(g).all(), k
        rounded += int(torch.equal(g.to(torch.bfloat16).float(), g))
    assert rounded == len(r["grads"]), "every gradient went through the bf16 wire buffer"


def _graph_worker(rank, world, port, out):
    """Two ranks, each replaying a 3-segment GraphedStep: buckets completed by a backward segment must be launched
    BEFORE the last segment's graph is replayed (they overlap it on RCCL), and the ranks must stay in lock-step."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["FAVIT_DP_VERIFY"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("focused-attention-vit_amd")
    pkg.set_compute_dtype("bf16")
    torch.manual_seed(3)
    m = pkg.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=6,
                                                  num_heads=4, window_size=7, use_mhla=True).cuda().train()
    opt = pkg.train.FusedAdamW(pkg.train.param_groups(m, lr=1e-2), lr=1e-2, weight_decay=0.0, bucket_mb=0.05)
    x, y = _batch()
    lo = rank * 4
    xs, ys = x[lo:lo + 4].cuda(), y[lo:lo + 4].cuda()
    step = pkg.train.GraphedStep(m, opt, xs, ys, segments=3)
    assert len(step.graphs) == 4
    # log (bucket, number of backward graphs replayed so far) at every launch
    replayed = {"n": 0}
    log = []

    class Proxy:
        def __init__(self, g, backward):
            self.g, self.backward = g, backward

        def replay(self):
            self.g.replay()
            if self.backward:
                replayed["n"] += 1
    step.graphs = [Proxy(g, i > 0) for i, g in enumerate(step.graphs)]
    for g_ in opt.groups:
        sync = g_["sync"]
        orig = sync._launch

        def wrapped(b, _orig=orig, _sync=sync):
            if not _sync._launched[b]:
                log.append((b, replayed["n"]))
            _orig(b)
        sync._launch = wrapped
    losses = []
    for _ in range(2):
        replayed["n"] = 0
        losses.append(float(step(xs, ys)))
    torch.cuda.synchronize()
    w = torch.cat([p.detach().flatten().cpu() for p in m.parameters()])
    verified = sum(getattr(g_["sync"], "verified", 0) for g_ in opt.groups)
    torch.save({"w": w, "log": log, "losses": losses, "verified": verified,
                "n_buckets": sum(len(g_["sync"].buckets) for g_ in opt.groups)}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_dp_graphed_step_launches_buckets_between_backward_segments(favit, tmp_path):
    out = str(tmp_path / "gdp")
    mp.spawn(_graph_worker, args=(2, 35500 + (os.getpid() % 2000), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0", weights_only=True), torch.load(out + ".1", weights_only=True)
    assert torch.equal(r0["w"], r1["w"]), "the two ranks diverged"
    for r in (r0, r1):
        early = [b for b, n in r["log"] if n < 3]
        assert len(r["log"]) == 2 * r["n_buckets"], r["log"]
        assert len(early) >= 2, f"no bucket went out before the last backward segment: {r['log']}"
        assert r["verified"] == 2 * r["n_buckets"]
    # same trajectory as one process training on the whole batch (mean over 8 = mean of the two rank means)
    favit.set_compute_dtype("bf16")
    try:
        torch.manual_seed(3)
        m = favit.models.vit_mhla.VisionTransformerMHLA(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=6,
                                                        num_heads=4, window_size=7, use_mhla=True).cuda().train()
        opt = favit.train.FusedAdamW(favit.train.param_groups(m, lr=1e-2), lr=1e-2, weight_decay=0.0, distributed=False)
        x, y = _batch()
        for _ in range(2):
            favit.train.train_step(m, x.cuda(), y.cuda(), opt)
        w = torch.cat([p.detach().flatten().cpu() for p in m.parameters()])
        assert (r0["w"] - w).norm() / w.norm() < 2e-3
    finally:
        favit.set_compute_dtype("fp32")
        favit.functional.clear_lp_mirrors()
