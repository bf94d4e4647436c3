"""Training-step harness: the counterpart of the reference's hot loop
(experiments/mhla_pretrained.py:350-372: zero_grad -> model(images) -> CrossEntropyLoss ->
backward -> AdamW.step) with the name-based parameter groups of
experiments/mhla_pretrained.py:308-327.  Loss and optimizer run in libfavit kernels
(favit_cross_entropy, favit_adamw); nothing in the step synchronises with the host.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import functional as F
from . import kernels as K
from .dp import FlatBuffers, GradSync


class _CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() (mean reduction) forward + gradient in one kernel."""

    @staticmethod
    def forward(ctx, logits, labels):
        B = logits.shape[0]
        rows, dlog = K.cross_entropy(logits.contiguous(), labels.contiguous(), grad_scale=1.0 / B)
        ctx.save_for_backward(dlog)
        return K.reduce_rows(rows.view(B, 1))[0] / B

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        return dlog * g, None


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    return _CrossEntropyFn.apply(logits, labels)


def param_groups(model: torch.nn.Module, lr: float, head_lr: Optional[float] = None,
                 latent_lr_mult: float = 5.0) -> List[Dict]:
    """The reference's name-based groups (experiments/mhla_pretrained.py:320-327): every
    trainable parameter whose name contains 'latent_proj' trains at 5x lr, 'head' at head_lr."""
    base, lat, head = [], [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if "head" in n:
            head.append(p)
        elif "latent_proj" in n:
            lat.append(p)
        else:
            base.append(p)
    out = [{"params": base, "lr": lr}]
    if lat:
        out.append({"params": lat, "lr": lr * latent_lr_mult})
    if head:
        out.append({"params": head, "lr": head_lr if head_lr is not None else lr})
    return [g for g in out if g["params"]]


class FusedAdamW:
    """torch.optim.AdamW semantics; one favit_adamw launch per parameter group over flat
    parameter / gradient / moment buffers (and the DP all-reduce runs on the same flat
    gradient buffers, see dp.py)."""

    def __init__(self, groups, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, bucket_mb=None,
                 distributed=None, wire_dtype=None):
        if isinstance(groups, torch.nn.Module):
            groups = [{"params": list(groups.parameters())}]
        elif groups and not isinstance(groups[0], dict):
            groups = [{"params": list(groups)}]
        self.groups = []
        dist_on = torch.distributed.is_initialized() if distributed is None else distributed
        for g in groups:
            flat = FlatBuffers(g["params"])
            self.groups.append({
                "flat": flat, "lr": g.get("lr", lr), "betas": g.get("betas", betas), "eps": g.get("eps", eps),
                "weight_decay": g.get("weight_decay", weight_decay),
                "m": torch.zeros_like(flat.flat_p), "v": torch.zeros_like(flat.flat_p),
                "lp": torch.empty(flat.numel, dtype=torch.bfloat16, device=flat.flat_p.device) if flat.flat_p.is_cuda else None,
                "sync": GradSync(flat, bucket_mb, wire_dtype=wire_dtype) if dist_on else None,
            })
            self.groups[-1]["mirror"] = None
            if self.groups[-1]["lp"] is not None:
                K.cast(flat.flat_p, torch.bfloat16, out=self.groups[-1]["lp"])
                self.groups[-1]["mirror"] = F.register_lp_mirror(flat.flat_p, self.groups[-1]["lp"], flat.params)
        self.steps = 0
        self.world = torch.distributed.get_world_size() if dist_on else 1
        syncs = [g["sync"] for g in self.groups if g["sync"] is not None]
        if syncs:
            F.set_grad_ready_hook(lambda p: [s.grad_ready(p) for s in syncs])

    def no_sync(self):
        """Gradient accumulation under data parallelism: ``with opt.no_sync(): loss.backward()`` for every
        micro-batch but the last (dp.GradSync.no_sync)."""
        import contextlib
        stack = contextlib.ExitStack()
        for g in self.groups:
            if g["sync"] is not None:
                stack.enter_context(g["sync"].no_sync())
        return stack

    def zero_grad(self):
        for g in self.groups:
            if g["sync"] is not None and (g["sync"]._handles or any(g["sync"]._launched)):
                # the previous backward did not reach step() (it raised, or the caller skipped the step): its
                # all-reduces may still be reading the gradient buffer this call is about to zero
                g["sync"].abort()
            g["flat"].zero_grad()
            # The bf16 mirror is rewritten by the AdamW kernel and validated per parameter at every use
            # (functional._LPMirror), so eager steps need no refresh here.  A captured step cannot run that host-side
            # check at replay time: GraphedStep.__call__ runs it before every replay (refresh_mirrors) -- round 3
            # recorded a full re-cast of every group in the graph instead (52 us of a 2.5-ms cfg3 step).

    def refresh_mirrors(self) -> None:
        """Re-cast the bf16 mirror of any group whose parameters were edited outside the optimizer since the mirror
        was last written (load_state_dict, nn.init, p.copy_, invalidate_weight_cache)."""
        for g in self.groups:
            if g["mirror"] is not None:
                g["mirror"].refresh_if_stale()

    def step(self):
        self.steps += 1
        for g in self.groups:
            if g["sync"] is not None:
                g["sync"].finish(average=False)
            f = g["flat"]
            K.adamw(f.flat_p, f.flat_g, g["m"], g["v"], g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                    g["weight_decay"], self.steps, grad_scale=1.0 / self.world, p_lp=g["lp"])
        # the AdamW kernel rewrote the bf16 mirrors itself: they stay valid across the epoch bump
        F.bump_weight_epoch([g["mirror"] for g in self.groups if g["mirror"] is not None])


class Health:
    """The library's health word (include/favit.h: favit_set_health_word): four device counters in which the
    cross-entropy and AdamW kernels note the FIRST non-finite loss row / gradient / updated parameter they meet, at no
    cost in a clean run.  `poll()` is a host sync; call it every N steps, or once after a timed region.  With
    `opt`, `report()` also walks the optimizer's flat buffers and names the first non-finite tensor."""

    def __init__(self, device=None):
        from . import _abi
        self.words = torch.zeros(4, dtype=torch.int32, device=device if device is not None else torch.cuda.current_device())
        _abi.check(_abi.lib().favit_set_health_word(self.words.data_ptr()), "favit_set_health_word")

    def close(self):
        from . import _abi
        _abi.check(_abi.lib().favit_set_health_word(None), "favit_set_health_word")

    def poll(self):
        """None while everything was finite, else a dict (kinds, first AdamW launch index, launches so far)."""
        f, first_loss, first_grad, launches = (int(v) & 0xFFFFFFFF for v in self.words.tolist())
        if f == 0:
            return None
        kinds = [k for b, k in ((1, "loss"), (2, "gradient"), (4, "parameter")) if f & b]
        return {"non_finite": kinds, "adamw_launch_of_first_bad_loss": first_loss or None,
                "adamw_launch_of_first_bad_gradient_or_parameter": first_grad or None, "adamw_launches": launches}

    def report(self, opt: "FusedAdamW" = None, model: torch.nn.Module = None, extra=()):
        """poll() plus, per optimizer group, the first non-finite entry of flat_p / flat_g / m / v mapped back to a
        parameter name (and of any (name, tensor) in `extra`)."""
        out = self.poll() or {}
        names = {}
        if model is not None:
            names = {id(p): n for n, p in model.named_parameters()}
        found = []
        for gi, g in enumerate(opt.groups if opt is not None else ()):
            flat = g["flat"]
            for key, t in (("flat_g", flat.flat_g), ("flat_p", flat.flat_p), ("m", g["m"]), ("v", g["v"])):
                bad = (~torch.isfinite(t)).nonzero()
                if bad.numel():
                    idx = int(bad[0])
                    who = "?"
                    for p_, o in zip(flat.params, flat.offsets):
                        if o <= idx < o + p_.numel():
                            who = names.get(id(p_), f"param@{o}")
                            break
                    found.append({"group": gi, "buffer": key, "count": int(bad.shape[0]), "first_index": idx, "parameter": who})
        for n, t in extra:
            if t is not None and torch.is_tensor(t) and t.is_floating_point() and not bool(torch.isfinite(t).all()):
                found.append({"tensor": n, "count": int((~torch.isfinite(t)).sum())})
        if found:
            out["tensors"] = found
        return out or None


def train_step(model, images, labels, opt: FusedAdamW):
    """One step of the reference's hot loop; returns the (device) loss tensor, no host sync."""
    opt.zero_grad()
    logits = model(images)
    loss = cross_entropy(logits, labels)
    loss.backward()
    opt.step()
    return loss


class GraphedStep:
    """One training step -- zero_grad -> forward -> cross-entropy -> backward -- captured ONCE in HIP graphs and
    replayed per step; the gradient all-reduce and the fused AdamW run eagerly.

    For the small-token configurations (SPPP+MHLA: 17 tokens per image) a step is ~600 kernel launches of a few
    microseconds each and the Python launch path, not the GPU, sets the step time; replayed graphs remove that.

    * Dropout (the reference trains with 0.1, main.py:106): seeds are kernel ARGUMENTS and are frozen into the graph,
      so a device "epoch" word is registered (functional.set_dropout_epoch) that every dropout-drawing kernel mixes
      into its seed when it EXECUTES; the first captured node increments it, i.e. every replay draws fresh masks and the
      forward and backward kernels of one replay agree.  An eager step with the same by-value seeds and the same epoch
      value draws exactly the masks of the replay (tests/test_gpu_modules.py).
    * Data parallelism: with ``segments`` > 1 the backward is captured as that many graphs (the block stack is split
      into consecutive autograd nodes, models/vit.py::run_encoder); after launching segment s the host reports the
      parameters whose gradients segment s completed to dp.GradSync, which puts every finished bucket on the wire
      while segment s + 1 replays.  ``segments = 1`` (default without a process group) defers every bucket to step().

    Requirements: FusedAdamW (its bf16 weight mirror is refreshed by kernels, not by host-side caching), fixed shapes,
    and for SPPP models ``model.assume_num_tokens`` set (the per-forward token-count check is a host sync)."""

    def __init__(self, model: torch.nn.Module, opt: FusedAdamW, images: torch.Tensor, labels: torch.Tensor,
                 warmup: int = 3, segments: Optional[int] = None, static_inputs: bool = False):
        """static_inputs: `images` / `labels` themselves are the buffers the captured kernels read (no clone at capture,
        no copy per call when the step is called with these same tensors): for a producer that writes every batch into
        fixed device buffers (bench.py's resident synthetic batch).  Default: private copies, one device-to-device copy
        of the batch per call."""
        if K.GEMM_TRACE is not None:
            raise RuntimeError("GraphedStep: disable kernels.GEMM_TRACE (event records cannot be captured)")
        self.model, self.opt = model, opt
        self.x, self.y = (images, labels) if static_inputs else (images.clone(), labels.clone())
        syncs = [g["sync"] for g in opt.groups if g["sync"] is not None and g["sync"]._active]
        self._syncs = syncs
        if segments is None:
            segments = 3 if syncs else 1
        self.segments = max(1, int(segments))
        for s in syncs:
            s.defer = True                     # nothing goes out while capturing / replaying: launches are explicit
        uses_dropout = model.training and any(isinstance(m, torch.nn.Dropout) and m.p > 0 for m in model.modules())
        self.epoch = None
        if uses_dropout:
            self.epoch = F.get_dropout_epoch()
            if self.epoch is None:
                self.epoch = torch.zeros(1, dtype=torch.int64, device=images.device)
                F.set_dropout_epoch(self.epoch)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up off the capture stream (lazy kernel attributes, allocator)
            for _ in range(max(1, warmup)):
                self._forward_backward(None)
        torch.cuda.current_stream().wait_stream(side)
        # capture: graph 0 = epoch bump + zero_grad + forward + loss; graphs 1..S = the backward pieces, output side
        # first.  One memory pool: activations saved by graph 0 are read by the backward graphs.
        prev_hook = F._STATE["grad_ready"]
        self._ready: List[List[torch.nn.Parameter]] = []
        self.graphs: List[torch.cuda.CUDAGraph] = []
        try:
            # (one graph and no process group: nobody consumes the notifications, and a registered hook makes the
            # encoder backward flush its batched parameter-gradient launches every four blocks instead of once)
            F.set_grad_ready_hook((lambda p: self._ready[-1].append(p)) if (syncs or self.segments > 1) else None)
            self._forward_backward(self.graphs)
        finally:
            F.set_grad_ready_hook(prev_hook)
        # which parameters are complete after which backward graph: the directly written ones reported themselves;
        # the others (gradients handed to autograd: cls_token, pos_embed, ...) are counted with the last piece
        seen = {id(p) for lst in self._ready for p in lst}
        self._ready[-1].extend(p for g_ in opt.groups for p in g_["flat"].params if id(p) not in seen)
        # Everything allocated OUTSIDE the graphs' pool whose address the captured kernels hold stays alive with the step:
        # the optimizer's flat buffers and bf16 mirrors (self.opt), the dropout epoch and the inputs (self.epoch / x / y),
        # the split-K slab workspace (kernels.py keeps outgrown ones) -- and the installed label maps (their content is
        # changed with SuperpixelSegmentation.update_label_maps; data.DeviceLoader does so once a step is captured).
        self._keep = [K._GROUPED_WS.get(self.x.device.index)]
        self._map_states = []          # label maps + the tensors derived from them (models/sppp.py::_MapState)
        for mod in model.modules():
            seg = getattr(mod, "segmentation", None)
            if seg is not None and getattr(seg, "_state", None) is not None:
                seg._captured = True
                self._map_states.append(seg._state)

    def _forward_backward(self, graphs):
        """zero_grad + forward + loss, then backward piece by piece; with a list, every piece is captured in a graph of
        its own (appended to it), without one it simply runs (warm-up)."""
        import contextlib

        def scope():
            if graphs is None:
                return contextlib.nullcontext()
            g = torch.cuda.CUDAGraph()
            pool = graphs[0].pool() if graphs else None
            graphs.append(g)
            return torch.cuda.graph(g, pool=pool) if pool is not None else torch.cuda.graph(g)

        with scope():
            if self.epoch is not None:
                self.epoch.add_(1)
            self.opt.zero_grad()
            with F.encoder_segments(self.segments) as seg:
                loss = cross_entropy(self.model(self.x), self.y)
            bounds = list(seg.boundaries)
        if graphs is not None:
            self.loss = loss
        heads, grads = [loss], [None]
        for k in range(len(bounds), -1, -1):               # the piece that starts at leaf k-1 (k = 0: the input side)
            if graphs is not None:
                self._ready.append([])
            with scope():
                torch.autograd.backward(heads, grads)
            if k > 0:
                out, leaf = bounds[k - 1]
                heads, grads = [out], [leaf.grad]

    def __call__(self, images: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        if images is not self.x:
            self.x.copy_(images, non_blocking=True)
        if labels is not self.y:
            self.y.copy_(labels, non_blocking=True)
        self.opt.refresh_mirrors()         # (host-side version check; a cast only after an outside edit of the weights)
        for st in self._map_states:        # (likewise: label maps edited in place without update_label_maps)
            st.refresh_if_stale()
        self.graphs[0].replay()
        for g, ready in zip(self.graphs[1:], self._ready):
            g.replay()
            if self._syncs and self.segments > 1:
                for s in self._syncs:
                    s.defer = False
                    for p in ready:
                        s.grad_ready(p)            # buckets completed by this segment go out under the next one
                    s.defer = True
        self.opt.step()            # eager: launches whatever is still deferred, waits, then AdamW
        return self.loss
