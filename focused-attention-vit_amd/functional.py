"""Forward/backward chains of the encoder hot path, built on the HIP kernels.

Each ``*Op`` below restates one reference forward (cited per class) as a sequence of
libfavit kernel launches and supplies the hand-written backward sequence.  ``OpFn`` is
the single ``torch.autograd.Function`` that plugs an Op into autograd so that the
``nn.Module`` mirrors in ``models/`` stay drop-in (``loss.backward()`` works unchanged).

Data layout in HBM: the residual stream is fp32 ``[B*L, D]``; every GEMM
operand is a contiguous row-major ``[rows, features]`` matrix in the compute dtype (bf16
or fp32); q, k~, v~ live interleaved in one ``[B*L, 3D]`` buffer exactly as the fused qkv
projection writes them; gradients of the residual stream are kept in fp32 plus a
compute-dtype copy that feeds the backward GEMMs.
"""
from __future__ import annotations

import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch

from . import kernels as K
from ._abi import ACT_DGELU, ACT_GELU, ACT_GELU_SAVEGRAD, ACT_MULAUX, ACT_NONE

# ------------------------------------------------------------------------------------
# configuration
# ------------------------------------------------------------------------------------
_STATE = {"cdt": torch.float32, "epoch": 0, "direct_grads": True, "grad_ready": None}


def set_direct_grads(on: bool) -> None:
    """When on (default), parameter gradients are accumulated by the kernels straight into an
    existing fp32 ``param.grad`` buffer (split-K atomics / fused reductions write there anyway),
    and autograd gets ``None`` for those parameters: no temporary gradient tensors, no zero-fill,
    no ``grad += g`` pass.  Parameters without a ``.grad`` buffer get ordinary returned grads."""
    _STATE["direct_grads"] = bool(on)


def set_grad_ready_hook(fn) -> None:
    """fn(param) is called after a parameter's gradient was accumulated directly (dp.GradSync)."""
    _STATE["grad_ready"] = fn


def _gt(p):
    """The gradient buffer of parameter p that kernels may accumulate into, or None."""
    if p is None or not _STATE["direct_grads"] or not getattr(p, "requires_grad", False):
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape or not g.is_cuda:
        return None
    return g


def _ready(*params):
    fn = _STATE["grad_ready"]
    if fn is not None:
        for p in params:
            if p is not None:
                if _SIDE["active"]:
                    _SIDE["pending"].append(p)       # gradient lives on the side stream: notify after the join
                else:
                    fn(p)


# ------------------------------------------------------------------------------------
# Optional side stream for weight-gradient GEMMs (dW = dY^T.X does not feed the backward chain).
# Measured on MI355X it bought nothing for this workload (21.42 vs 21.35 ms/step: every GEMM already
# fills the chip), so it is OFF by default; kept because it is the natural hook for overlapping the
# DP all-reduce of early buckets with compute on multi-GPU runs.  Also measured: the small latency-bound
# latent_proj fold kernels (36 per step, 15-40 us) moved to this stream (forward folds launched up
# front, backward folds under the LayerNorm backward) -- no gain either (18.7 vs 18.3 ms/step): every
# cross-stream dependency costs a barrier packet of several microseconds, about what the overlap saves.
# ------------------------------------------------------------------------------------
_SIDE = {"enabled": False, "stream": None, "active": False, "pending": [], "used": False}


def set_side_stream(on: bool) -> None:
    _SIDE["enabled"] = bool(on)


class _side_stream:
    """with _side_stream(t1, t2, ...): launches inside run on the side stream after everything
    already queued on the current stream; the tensors are marked as used by it."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]

    def __enter__(self):
        if not _SIDE["enabled"]:
            self.ctx = None
            return self
        if _SIDE["stream"] is None:
            _SIDE["stream"] = torch.cuda.Stream()
        s = _SIDE["stream"]
        s.wait_stream(torch.cuda.current_stream())
        for t in self.tensors:
            t.record_stream(s)
        self.ctx = torch.cuda.stream(s)
        self.ctx.__enter__()
        _SIDE["active"] = True
        _SIDE["used"] = True
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            _SIDE["active"] = False
            self.ctx.__exit__(*exc)
        return False


def join_side_stream() -> None:
    """Current stream waits for the side stream; deferred gradient-ready notifications fire."""
    if _SIDE["used"]:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])
        _SIDE["used"] = False
    if _SIDE["pending"]:
        fn = _STATE["grad_ready"]
        pend, _SIDE["pending"] = _SIDE["pending"], []
        if fn is not None:
            for p in pend:
                fn(p)


def set_compute_dtype(d) -> None:
    """'fp32' (exact f32 MFMA path, the parity mode), 'bf16' (bf16 MFMA, fp32 accumulate) or 'fp8'
    (BASELINE.json configs[3]: the nn.Linear GEMMs of the encoder blocks run on fp8 MFMA -- e4m3
    activations / weights, e5m2 gradients, per-tensor scales taken on the device -- everything else as
    in 'bf16'; patch embedding and classifier head stay bf16)."""
    fp8 = False
    if isinstance(d, str):
        fp8 = d == "fp8"
        d = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
             "fp8": torch.bfloat16}[d]
    if d not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be fp32, bf16 or fp8")
    _STATE["cdt"] = d
    _STATE["fp8"] = fp8


def get_compute_mode() -> str:
    return "fp8" if _STATE.get("fp8") else ("bf16" if _STATE["cdt"] == torch.bfloat16 else "fp32")


def get_compute_dtype() -> torch.dtype:
    return _STATE["cdt"]


def bump_weight_epoch(synced_mirrors=()) -> None:
    """Called by optimizers that update parameters through raw pointers.  synced_mirrors: the bf16
    mirrors the caller has just rewritten itself (they stay valid across the bump)."""
    _STATE["epoch"] += 1
    for m in synced_mirrors:
        m.epoch = _STATE["epoch"]


# parameter storage -> (fp32 flat buffer, bf16 mirror) kept fresh by the fused optimizer (train.FusedAdamW)
class _LPMirror:
    """bf16 mirror ``lp`` of the flat fp32 parameter buffer ``flat_p`` of one optimizer group.

    The owner (train.FusedAdamW) rewrites it inside the AdamW kernel and tells the mirror so
    (``mark_synced``).  Every other way of editing a parameter is detected when the parameter is next
    used: ``load_state_dict`` / ``nn.init.*`` / ``p.copy_`` bump the tensor's version counter (compared
    per parameter), ``invalidate_weight_cache()`` bumps the weight epoch (compared per mirror; needed
    for ``p.data.copy_``, which no counter sees).  A stale slice / mirror is re-cast before it is returned."""
    __slots__ = ("flat_ref", "lp", "epoch", "versions", "params")

    def __init__(self, flat_p, lp, params=()):
        self.flat_ref, self.lp, self.params = weakref.ref(flat_p), lp, [weakref.ref(p) for p in params]
        self.mark_synced()

    def mark_synced(self):
        """The owner has just rewritten the whole mirror from flat_p (AdamW kernel or a cast): every
        parameter's current version counter is the one the mirror reflects."""
        self.epoch = _STATE["epoch"]
        self.versions = {id(p): p._version for p in (r() for r in self.params) if p is not None}

    def refresh_if_stale(self) -> bool:
        """Host-side check for callers that cannot check at use time (a replayed HIP graph reads the mirror without
        going through lookup): re-cast the whole mirror if any parameter was edited since it was last written."""
        fp = self.flat_ref()
        if fp is None:
            return False
        stale = self.epoch != _STATE["epoch"]
        if not stale:
            for r in self.params:
                p = r()
                if p is not None and self.versions.get(id(p), p._version) != p._version:
                    stale = True
                    break
        if stale:
            K.cast(fp, torch.bfloat16, out=self.lp)
            self.mark_synced()
        return stale

    def lookup(self, src, w):
        fp = self.flat_ref()
        if fp is None:
            return None
        base, ptr = fp.data_ptr(), w.data_ptr()
        if not (base <= ptr < base + 4 * fp.numel()):
            return False
        if self.epoch != _STATE["epoch"]:                 # explicit invalidation: refresh the whole mirror once
            K.cast(fp, torch.bfloat16, out=self.lp)
            self.epoch = _STATE["epoch"]
        off = (ptr - base) // 4
        view = self.lp[off:off + w.numel()].view(w.shape)
        key = id(src)
        ver = src._version
        if self.versions.get(key, ver) != ver:            # edited in place since the last use: re-cast this slice
            K.cast(w, torch.bfloat16, out=view)
        self.versions[key] = ver
        return view


_LP_MIRRORS: List[_LPMirror] = []


def register_lp_mirror(flat_p: torch.Tensor, flat_lp: torch.Tensor, params=()) -> _LPMirror:
    """flat_lp is a bf16 mirror of the fp32 parameter buffer flat_p, just cast from it; params are the
    nn.Parameters living in flat_p (see _LPMirror)."""
    m = _LPMirror(flat_p, flat_lp, params)
    _LP_MIRRORS.append(m)
    return m


def clear_lp_mirrors() -> None:
    _LP_MIRRORS.clear()


def invalidate_weight_cache() -> None:
    """Drop every cached compute-dtype copy of a parameter (per-tensor caches and the fused optimizer's
    bf16 mirrors alike).  Needed only after editing parameters through ``.data`` (``p.data.copy_(...)``),
    which autograd's version counter cannot see; ``load_state_dict``, ``nn.init.*``, ``p.copy_`` and
    optimizer steps are detected automatically."""
    _STATE["epoch"] += 1


def wcast(w: torch.Tensor) -> torch.Tensor:
    """Parameter in the compute dtype.  The copy is cached ON the tensor object that was passed in
    (so it dies with the parameter: another model whose weights later reuse the same address can
    never see it) and is tagged with (address, version counter, weight epoch, dtype)."""
    cdt = _STATE["cdt"]
    src = w
    w = w.detach()
    if w.dtype == cdt and w.is_contiguous():
        return w
    if cdt == torch.bfloat16 and w.dtype == torch.float32 and w.is_contiguous():
        for m in list(_LP_MIRRORS):
            hit = m.lookup(src, w)
            if hit is None:                   # the optimizer that owned this mirror is gone
                _LP_MIRRORS.remove(m)
            elif hit is not False:
                return hit
    tag = (w.data_ptr(), tuple(w.shape), w._version, _STATE["epoch"], cdt)
    hit = getattr(src, "_favit_cast", None)
    if hit is not None and hit[0] == tag:
        return hit[1]
    c = K.cast(w, cdt)
    try:
        src._favit_cast = (tag, c)
    except AttributeError:                    # pragma: no cover  (objects without a __dict__)
        pass
    return c


def set_dropout_epoch(word: Optional[torch.Tensor]) -> None:
    """Register (or, with None, clear) the device word every dropout-drawing kernel mixes into its seed at execution
    time (favit_set_dropout_epoch): a one-element int64 tensor on the current device that the caller increments once per
    step.  With it, dropout seeds frozen into a captured HIP graph still give fresh masks at every replay."""
    if word is not None:
        K.require_gpu(word)
        if word.dtype != torch.int64 or word.numel() != 1:
            raise TypeError("the dropout epoch is a one-element int64 device tensor")
    _STATE["drop_epoch"] = word
    from . import _abi
    _abi.check(_abi.lib().favit_set_dropout_epoch(None if word is None else word.data_ptr()), "favit_set_dropout_epoch")


def get_dropout_epoch() -> Optional[torch.Tensor]:
    return _STATE.get("drop_epoch")


def _seed() -> int:
    # drawn from torch's CPU generator: reproducible under torch.manual_seed, no device sync
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing() and _STATE.get("drop_epoch") is None:
        raise RuntimeError("dropout inside a captured HIP graph: the seed is a kernel argument and would be frozen "
                           "into the graph; register a dropout epoch word first (functional.set_dropout_epoch; "
                           "train.GraphedStep does it for models with dropout)")
    return int(torch.empty((), dtype=torch.int64).random_().item())


# Backward in segments (train.GraphedStep): models/vit.py::run_encoder splits the block stack into this many autograd
# nodes and reports the tensors between them.
_SEG = {"n": 1, "boundaries": []}


class encoder_segments:
    """with encoder_segments(n): forward passes split every encoder into n autograd nodes that are CUT APART: node k + 1
    reads a detached leaf copy of node k's output.  .boundaries collects the (output of node k, leaf input of node
    k + 1) pairs in forward order; backward then runs back to front, one autograd call per piece:
    backward(loss); backward(out_k, grad_tensors=in_{k+1}.grad) ... (train.GraphedStep)."""

    def __init__(self, n: int):
        self.n = max(1, int(n))

    def __enter__(self):
        self.prev = (_SEG["n"], _SEG["boundaries"])
        _SEG["n"], _SEG["boundaries"] = self.n, []
        self.boundaries = _SEG["boundaries"]
        return self

    def __exit__(self, *exc):
        _SEG["n"], _SEG["boundaries"] = self.prev
        return False


def get_encoder_segments() -> int:
    return _SEG["n"]


def note_segment_boundary(x: torch.Tensor) -> torch.Tensor:
    leaf = x.detach().requires_grad_(True)
    _SEG["boundaries"].append((x, leaf))
    return leaf


def _as_cdt(x: torch.Tensor) -> torch.Tensor:
    cdt = _STATE["cdt"]
    x = x.contiguous()
    return x if x.dtype == cdt else K.cast(x, cdt)


def _as_f32(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous()
    return x if x.dtype == torch.float32 else K.cast(x, torch.float32)


def _mask_u8(mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if mask is None:
        return None
    return (mask != 0).to(torch.uint8).contiguous()


# ------------------------------------------------------------------------------------
# GEMM helpers (nn.Linear forward / backward)
# ------------------------------------------------------------------------------------
E4M3, E5M2 = torch.float8_e4m3fn, torch.float8_e5m2


# Delayed scaling (fp8 mode): one amax history per quantisation site = (weight nn.Parameter of the Linear, operand
# role).  Callers that pass no site measure the amax of every tensor in a separate pass (as before).
_FP8_HIST = {}


def _fp8_hist(site, role: str):
    """site: a long-lived object that identifies the Linear (its weight nn.Parameter); None = no history."""
    if site is None or torch.cuda.is_current_stream_capturing():      # the slot rotation is host state
        return None
    key = (id(site), role)
    ent = _FP8_HIST.get(key)
    if ent is None or ent[0]() is not site:           # (weakref guards against a recycled id)
        ent = (weakref.ref(site), K.Fp8History(site.device))
        _FP8_HIST[key] = ent
        if len(_FP8_HIST) > 4096:                     # models come and go in a test process
            for k in [k for k, v in _FP8_HIST.items() if v[0]() is None]:
                del _FP8_HIST[k]
    return ent[1]


def _q8(t: torch.Tensor, fmt: torch.dtype, want_t: bool = False, hist=None):
    """(q, q_t or None, scale_inv) of a 2-D bf16 matrix, cached on the tensor object.  Activations and gradients are
    only needed row-major (forward and input-gradient GEMMs); the transposed copy is for weights (W^T of the
    input-gradient GEMM).  Weight-gradient GEMMs stay on the bf16 grouped launch: measured at ViT-Base, the fp8
    weight-gradient GEMMs (split-K with fp32 atomics, both operands quantised AND transposed first) cost 9.6 ms per
    step against 7.5 ms for the bf16 grouped launch + its reduction, before counting the transposing passes."""
    hit = getattr(t, "_favit_q8", None)
    if hit is not None and hit[0] == (fmt, t._version) and (hit[1][1] is not None or not want_t):
        return hit[1]
    q, qt, sinv = K.fp8_quantize(t, fmt, want=True, want_t=want_t, hist=hist)
    out = (q, qt, sinv)
    try:
        t._favit_q8 = ((fmt, t._version), out)
    except AttributeError:                    # pragma: no cover
        pass
    return out


def _q8_request(chain, prm, which: str, D: int):
    """fp8 mode: (fp8 dtype, amax history) of the GEMM operand that the tensor a LayerNorm pass is about to write will
    become -- `which` = "fwd": the chain's input (qkv / fc1 activations, e4m3), "bwd": the gradient of its output (the
    proj / fc2 input-gradient GEMM's dY, e5m2).  K.layernorm_fwd / _bwd then write the quantised copy in the same pass
    and park it where _q8 finds it (favit_layernorm_*_q8).  None: not fp8 mode, an unknown chain, a contraction the
    fp8 GEMM does not take, or no history (graph capture)."""
    if not _STATE.get("fp8") or get_compute_dtype() != torch.bfloat16 or D % 64 != 0 or os.environ.get("FAVIT_FP8_NO_LNQ8"):
        return None
    idx = getattr(chain, "q8_" + which, None)
    if idx is None or prm is None or idx >= len(prm) or prm[idx] is None:
        return None
    hist = _fp8_hist(prm[idx], "a" if which == "fwd" else "dy")
    return None if hist is None else ((E4M3 if which == "fwd" else E5M2), hist)


def _use_fp8(allow, *mats, k_dims=()):
    """fp8 GEMM applies: fp8 mode, bf16 row-major operands, contraction lengths multiples of 64."""
    if not (allow and _STATE.get("fp8")):
        return False
    return all(m.dtype == torch.bfloat16 and m.dim() == 2 and m.is_contiguous() for m in mats) and \
        all(k % 64 == 0 for k in k_dims)


def lin_fwd(a, w_c, bias, M, N, Kd, out_dtype, *, act=ACT_NONE, residual=None, want_pre=False, drop=(0.0, 0),
            allow_fp8=True, site=None):
    out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    pre = torch.empty((M, N), dtype=out_dtype, device=a.device) if want_pre else None
    if _use_fp8(allow_fp8, a, w_c, k_dims=(Kd,)):
        aq, _, sa = _q8(a, E4M3, hist=_fp8_hist(site, "a"))
        wq, _, sw = _q8(w_c, E4M3, want_t=True, hist=_fp8_hist(site, "w"))      # the backward wants W^T from the same pass
        K.gemm(aq, wq, out, M, N, Kd, Kd, Kd, N, bias=bias, act=act, aux_out=pre, ld_aux_out=N, residual=residual,
               ld_res=N, dropout_p=drop[0], dropout_seed=drop[1], scale_a=sa, scale_b=sw)
        return (out, pre) if want_pre else out
    K.gemm(a, w_c, out, M, N, Kd, Kd, Kd, N, bias=bias, act=act, aux_out=pre, ld_aux_out=N, residual=residual,
           ld_res=N, dropout_p=drop[0], dropout_seed=drop[1])
    return (out, pre) if want_pre else out


# LayerNorm fused into the qkv / fc1 projections at short token counts (favit_ln_gemm).  OFF by default: built, at
# parity (tests/test_gpu_kernels.py::test_layernorm_fused_into_small_gemm) and measured at no gain -- cfg3 2.53 ms
# fused vs 2.48 separate, cfg1 1.90 vs 1.92, cfg5 2.13 vs 2.07: the normalisation is a serial latency chain (load the
# rows, two reductions, normalise, write the LDS panel) in front of the first MFMA of every workgroup, about as long as
# the 4.9-us launch it replaces, and it is repeated by the 9-12 column tiles of a row block.  FAVIT_LN_FUSE=1 enables it.
_LN_FUSE = bool(os.environ.get("FAVIT_LN_FUSE"))


def lin_fwd_ln(ln, w_c, bias, M, N, Kd, out_dtype, *, act=ACT_NONE, want_pre=False, drop=(0.0, 0)):
    """lin_fwd whose A operand is LayerNorm(x): one launch at short token counts (favit_ln_gemm), else layernorm_fwd +
    lin_fwd.  ln = (x fp32 [M, Kd], gamma, beta).  Returns (out, pre or None, xn, mean, rstd)."""
    x, gamma, beta = ln
    if _LN_FUSE and w_c.dtype == torch.bfloat16 and get_compute_mode() != "fp8":
        out = torch.empty((M, N), dtype=out_dtype, device=x.device)
        pre = torch.empty((M, N), dtype=out_dtype, device=x.device) if want_pre else None
        r = K.ln_gemm(x, Kd, gamma, beta, w_c, out, M, N, Kd, bias=bias, act=act, aux_out=pre, dropout_p=drop[0],
                      dropout_seed=drop[1])
        if r is not None:
            return (out, pre) + r
    xn, mu, rs = K.layernorm_fwd(x, Kd, gamma, beta, M, Kd, get_compute_dtype())
    o = lin_fwd(xn, w_c, bias, M, N, Kd, out_dtype, act=act, want_pre=want_pre, drop=drop)
    return (o + (xn, mu, rs)) if want_pre else (o, None, xn, mu, rs)


def lin_bwd_x(dy, w_c, M, N, Kd, out_dtype, *, dgelu_pre=None, pre_is_grad=False, drop=(0.0, 0), allow_fp8=True,
              site=None):
    """dx[M,K] = dy[M,N] @ w[N,K]  (optionally * gelu'(pre) * dropout-mask; pre_is_grad: `dgelu_pre` already holds
    gelu'(pre), saved by the forward's ACT_GELU_SAVEGRAD epilogue)."""
    dx = torch.empty((M, Kd), dtype=out_dtype, device=dy.device)
    act = (ACT_MULAUX if pre_is_grad else ACT_DGELU) if dgelu_pre is not None else ACT_NONE
    if _use_fp8(allow_fp8, dy, w_c, k_dims=(N,)) and (dgelu_pre is None or dgelu_pre.dtype == torch.bfloat16):
        dyq, _, sdy = _q8(dy, E5M2, hist=_fp8_hist(site, "dy"))
        _, wqt, sw = _q8(w_c, E4M3, want_t=True, hist=_fp8_hist(site, "w"))      # (cached from the forward)
        K.gemm(dyq, wqt, dx, M, Kd, N, N, wqt.stride(0), Kd, act=act, aux_in=dgelu_pre, ld_aux_in=Kd,
               dropout_p=drop[0], dropout_seed=drop[1], scale_a=sdy, scale_b=sw)
        return dx
    K.gemm(dy, w_c, dx, M, Kd, N, N, Kd, Kd, b_kmajor=False, act=act,
           aux_in=dgelu_pre, ld_aux_in=Kd, dropout_p=drop[0], dropout_seed=drop[1])
    return dx


# Weight-gradient GEMMs are collected -- over SEVERAL blocks at short token counts -- and issued as one grouped launch
# (favit_gemm_grouped_tn): nothing downstream in the backward chain reads a weight gradient, so the launch can wait,
# and the more tiles it holds, the fewer K-splits it needs to fill the chip (round 4; csrc/gemm.hip grouped_plan: the
# twelve blocks of cfg3 in ONE launch without any split, slab or reduction kernel: 19 us per block against 40).
# How long it waits is bounded by the bytes the waiting operands keep alive (FAVIT_WGRAD_HOLD_MB, default 384: about
# the Infinity Cache + L2): at 50,432 tokens one block already holds 620 MB, so cfg2 / cfg4 launch per block as
# before -- measured, four cfg2 blocks per launch make the launch itself 2.5 % faster and the STEP 1 % slower
# (14.49 vs 14.35 ms; cfg4 28.99 vs 28.82): the held dY buffers are no longer recycled block after block, every
# input-gradient GEMM writes into memory that is cold in the Infinity Cache and the TLB (those GEMMs: 3.74 -> 3.84 ms).
# At most four blocks per launch when somebody listens for finished gradients (data parallelism: the buckets of those
# blocks go out while the rest of backward runs; the graph segments of train.GraphedStep end a launch as well).
_WG = {"list": None, "blocks": 0, "every": 1, "bytes": 0}
_WG_HOLD_BYTES = int(float(os.environ.get("FAVIT_WGRAD_HOLD_MB", "384")) * (1 << 20))


def begin_wgrads() -> None:
    every = 4 if _STATE["grad_ready"] is not None else K.GROUP_MAX // 4
    if _SIDE["enabled"] or os.environ.get("FAVIT_WGRAD_PER_BLOCK"):
        every = 1                             # (side-stream mode and the A/B switch: one launch per block, as in round 3)
    _WG.update(list=[], blocks=0, every=every, bytes=0)


def flush_wgrads(force: bool = True) -> bool:
    """Launch the collected weight-gradient GEMMs (grouped if the library can, else one by one).  force=False: called
    at the end of a block -- launches only every `every` blocks or once the waiting operands exceed the hold limit.
    Returns True if the list is empty afterwards."""
    lst = _WG["list"]
    if lst is None:
        return True
    if not force:
        _WG["blocks"] += 1
        nb = _WG["blocks"]
        # (would one more block of the same size still fit the hold limit and the launch?)
        if nb < _WG["every"] and _WG["bytes"] * (nb + 1) <= _WG_HOLD_BYTES * nb and len(lst) * (nb + 1) <= K.GROUP_MAX * nb:
            return not lst
    _WG["blocks"] = 0
    _WG["bytes"] = 0
    if lst:
        _WG["list"] = []
        probs = [(dy, a, dw, db, acc) for dy, a, dw, db, acc, _ in lst]
        with _side_stream(*[t for pr in probs for t in pr[:4]]):
            if not K.gemm_grouped_tn(probs):       # (one problem too: fine-tuning with frozen layers leaves only dWeff)
                for dy, a, dw, db, acc in probs:
                    K.gemm(dy, a, dw, dy.shape[1], a.shape[1], dy.shape[0], dy.stride(0), a.stride(0), dw.stride(0),
                           a_kmajor=False, b_kmajor=False, a_rowsum=db, accumulate=acc)
            for *_, ready in lst:
                _ready(*ready)
    return True


def end_wgrads() -> None:
    flush_wgrads()
    _WG["list"] = None


# Small latency-bound launches that only produce PARAMETER gradients (nothing downstream in the backward chain reads
# them) are collected per encoder backward and issued batched: the latent_proj fold backward (one ~100-workgroup launch
# per layer: 22.9 us x 12 per cfg2 step) and the fold of the LayerNorm dgamma / dbeta partial sums (one per LayerNorm
# backward: 4.7 us x 24).  Flushed every `every` blocks, so that under data parallelism the buckets that hold these
# gradients are still reduced while the rest of backward runs, and at the end of the encoder backward.
_DEFER = {"on": False, "fold": [], "ln": []}


def begin_deferred() -> None:
    # single process: one batch at the end of the encoder backward; under data parallelism whenever the grouped
    # weight-gradient launch has gone out (every 4 blocks), so that the buckets holding these gradients start their
    # all-reduce while the rest of backward still runs.  The fold consumes dWeff: never before that launch.
    _DEFER.update(on=not _SIDE["enabled"], fold=[], ln=[])


def flush_deferred() -> None:
    if not _DEFER["on"]:
        return
    assert not _WG["list"], "the latent_proj fold reads weight gradients that have not been launched"
    fold, ln = _DEFER["fold"], _DEFER["ln"]
    _DEFER["fold"], _DEFER["ln"] = [], []
    if ln:
        K.reduce_rows_multi([e[:3] for e in ln])
        for e in ln:
            _ready(*e[3])
    by_h = {}
    for e in fold:
        by_h.setdefault((e[6], tuple(e[2].shape), e[5][0] is None), []).append(e)
    for (H, _, _), es in by_h.items():
        K.mhla_fold_bwd_multi([e[:6] for e in es], H)
        for e in es:
            _ready(*e[7])


def end_deferred() -> None:
    flush_deferred()
    _DEFER["on"] = False


# Zero-initialised fp32 vectors for fused column sums that have no gradient buffer to accumulate into (the bias
# gradient of the folded qkv weights, one per block): carved out of ONE zero-filled pool per encoder backward instead
# of one 5-us fill launch each (12 per cfg2 step).
_ZPOOL = {"buf": None, "off": 0}


def begin_zero_pool(n_floats: int, device) -> None:
    _ZPOOL["buf"] = torch.zeros(n_floats, dtype=torch.float32, device=device) if n_floats > 0 else None
    _ZPOOL["off"] = 0


def end_zero_pool() -> None:
    _ZPOOL["buf"] = None


def _zero_vec(n: int, device) -> torch.Tensor:
    buf = _ZPOOL["buf"]
    n4 = (n + 3) // 4 * 4                               # keep every slice 16-byte aligned (slab reduction)
    if buf is not None and buf.device == device and _ZPOOL["off"] + n4 <= buf.numel():
        o = _ZPOOL["off"]
        _ZPOOL["off"] = o + n4
        return buf[o:o + n]
    return torch.zeros(n, dtype=torch.float32, device=device)


def lin_bwd_w(dy, a, M, N, Kd, want_bias=True, wp=None, bp=None, allow_fp8=True):
    """dw[N,K] = dy[M,N]^T @ a[M,K] (fp32, split-K over the tokens), db[N] = column sums of dy (fused).
    If the parameters wp / bp own usable .grad buffers the results are accumulated there and None
    is returned in their place."""
    frozen_w = wp is not None and not wp.requires_grad
    frozen_b = (not want_bias) or (bp is not None and not bp.requires_grad)
    if frozen_w and frozen_b:          # frozen layer (experiments/mhla_pretrained.py:237-247): no weight-gradient GEMM
        return None, None
    tw = _gt(wp)
    tb = _gt(bp) if want_bias else None
    dw = tw if tw is not None else torch.empty((N, Kd), dtype=torch.float32, device=dy.device)
    db = None
    if want_bias:
        db = tb if tb is not None else _zero_vec(N, dy.device)
    if _WG["list"] is not None and dy.dtype == torch.bfloat16:
        # deferred: joins the block's grouped weight-gradient launch
        _WG["list"].append((dy, a, dw, db, tw is not None,
                            [p for p, t in ((wp, tw), (bp, tb)) if t is not None]))
        _WG["bytes"] += dy.numel() * dy.element_size() + a.numel() * a.element_size()
        return (None if tw is not None else dw), (None if tb is not None else db)
    with _side_stream(dy, a, dw, db):
        # a long-token weight gradient outside a block (patch embedding: 50,176 tokens at cfg2): the grouped launch
        # with ONE problem -- K-splits into slabs + fixed-order reduction -- instead of split-K with fp32 atomics
        # (73 -> ~45 us, and bitwise reproducible); it declines what it cannot take (short or ragged token counts)
        if not (dy.dtype == torch.bfloat16 and M >= 8192 and (db is not None or not want_bias) and
                K.gemm_grouped_tn([(dy, a, dw, db, tw is not None)])):
            K.gemm(dy, a, dw, N, Kd, M, N, Kd, Kd, a_kmajor=False, b_kmajor=False, a_rowsum=db,
                   accumulate=tw is not None)
    if tw is not None:
        _ready(wp)
    if tb is not None:
        _ready(bp)
    return (None if tw is not None else dw), (None if tb is not None else db)


# ------------------------------------------------------------------------------------
# scaled-dot-product attention on the MFMA GEMM (dense variants)
# ------------------------------------------------------------------------------------
class _View:
    """[rows, ld] matrix holding per-(batch, head) [L, hd] blocks: element (b,h,l,d) at
    off + b*sb + h*sh + l*ld + d."""
    __slots__ = ("t", "off", "ld", "sb", "sh")

    def __init__(self, t, off, ld, sb, sh):
        self.t, self.off, self.ld, self.sb, self.sh = t, off, ld, sb, sh


def sdpa_fwd(q: _View, k: _View, v: _View, o: _View, B, H, Lq, Lk, hd, scale, mask, m_sb, m_sq, p, seed):
    """softmax(scale * q k^T, masked) (dropout) v on the fused kernel (csrc/sdpa.hip): no [B*H, Lq, Lk] tensor
    in HBM.  Returns what backward needs: (lse,) -- or, for a head dim the fused kernel does not take (not a
    multiple of 16), the unfused MFMA-GEMM + softmax chain's (P, Pd)."""
    if hd % 16 == 0:
        return (K.sdpa_fwd(q, k, v, o, B, H, Lq, Lk, hd, scale, mask, m_sb, m_sq, p, seed),)
    cdt = q.t.dtype
    Z = B * H
    S = torch.empty((Z, Lq, Lk), dtype=torch.float32, device=q.t.device)
    K.gemm(q.t, k.t, S, Lq, Lk, hd, q.ld, k.ld, Lk, alpha=scale, batch=Z, batch_inner=H, sA=(q.sb, q.sh),
           sB=(k.sb, k.sh), sC=(H * Lq * Lk, Lq * Lk), a_off=q.off, b_off=k.off)
    P, Pd = K.softmax_fwd(S, cdt, H, Z, Lq, Lk, mask, m_sb, m_sq, p, seed)
    K.gemm(Pd, v.t, o.t, Lq, hd, Lk, Lk, v.ld, o.ld, b_kmajor=False, batch=Z, batch_inner=H,
           sA=(H * Lq * Lk, Lq * Lk), sB=(v.sb, v.sh), sC=(o.sb, o.sh), b_off=v.off, c_off=o.off)
    return P, Pd


def sdpa_bwd(q: _View, k: _View, v: _View, o: _View, do: _View, dq: _View, dk: _View, dv: _View, saved, B, H, Lq, Lk,
             hd, scale, mask, m_sb, m_sq, p, seed):
    if len(saved) == 1:
        K.sdpa_bwd(q, k, v, o, do, dq, dk, dv, saved[0], B, H, Lq, Lk, hd, scale, mask, m_sb, m_sq, p, seed)
        return
    P, Pd = saved
    Z = B * H
    sP = (H * Lq * Lk, Lq * Lk)
    dPd = torch.empty((Z, Lq, Lk), dtype=torch.float32, device=q.t.device)
    K.gemm(do.t, v.t, dPd, Lq, Lk, hd, do.ld, v.ld, Lk, batch=Z, batch_inner=H, sA=(do.sb, do.sh), sB=(v.sb, v.sh),
           sC=sP, a_off=do.off, b_off=v.off)
    dS = K.softmax_bwd(P, dPd, Z, Lq, Lk, p, seed)
    # dQ = scale * dS . K
    K.gemm(dS, k.t, dq.t, Lq, hd, Lk, Lk, k.ld, dq.ld, b_kmajor=False, alpha=scale, batch=Z, batch_inner=H, sA=sP,
           sB=(k.sb, k.sh), sC=(dq.sb, dq.sh), b_off=k.off, c_off=dq.off)
    # dK = scale * dS^T . Q
    K.gemm(dS, q.t, dk.t, Lk, hd, Lq, Lk, q.ld, dk.ld, a_kmajor=False, b_kmajor=False, alpha=scale, batch=Z,
           batch_inner=H, sA=sP, sB=(q.sb, q.sh), sC=(dk.sb, dk.sh), b_off=q.off, c_off=dk.off)
    # dV = Pd^T . dO
    K.gemm(Pd, do.t, dv.t, Lk, hd, Lq, Lk, do.ld, dv.ld, a_kmajor=False, b_kmajor=False, batch=Z, batch_inner=H,
           sA=sP, sB=(do.sb, do.sh), sC=(dv.sb, dv.sh), b_off=do.off, c_off=dv.off)


# ------------------------------------------------------------------------------------
# attention chains: input = LayerNorm output xn [M, D] (compute dtype), output fp32 [M, D]
# (+ residual fused into the projection epilogue).  bwd takes the compute-dtype copy of the
# output gradient and returns (dxn [M,D] compute dtype, [param grads in `names` order]).
# ------------------------------------------------------------------------------------
class MHLAChain:
    q8_fwd, q8_bwd = 0, 4        # prm index of the weight whose GEMM consumes the chain's input (qkv) / output gradient (proj)
    """MultiHeadLatentAttention.forward (models/mhla.py:85-161)."""
    names = ("qkv.weight", "qkv.bias", "latent_proj.weight", "latent_proj.bias", "proj.weight", "proj.bias")

    def __init__(self, H, W, p_attn=0.0, p_proj=0.0):
        if W % 2 == 0:
            raise ValueError("window_size must be odd: the reference crashes on even sizes (models/mhla.py:83)")
        self.H, self.W, self.p_attn, self.p_proj = H, W, p_attn, p_proj

    def fwd(self, xn, prm, B, L, residual, mask, training, pre=None, ln=None):
        """ln = (x, gamma, beta, out): the LayerNorm in front of this branch is fused into the qkv projection (xn is
        None); out receives (xn, mean, rstd)."""
        wqkv, bqkv, wl, bl, wp, bp = prm
        M, D = xn.shape if ln is None else ln[0].shape
        cdt = xn.dtype if ln is None else get_compute_dtype()
        H, hd = self.H, D // self.H
        pa = self.p_attn if training else 0.0
        pp = self.p_proj if training else 0.0
        sa = _seed() if pa > 0 else 0
        sp = _seed() if pp > 0 else 0
        weff, beff = pre if pre is not None else K.mhla_fold_fwd(wqkv, bqkv, wl, bl, H, cdt)
        if ln is not None:
            qkv, _, xn, mu, rs = lin_fwd_ln(ln[:3], weff, beff, M, 3 * D, D, cdt)
            ln[3][:] = [xn, mu, rs]
        else:
            qkv = lin_fwd(xn, weff, beff, M, 3 * D, D, xn.dtype, site=wqkv)
        # training: the forward also leaves lse per row, and backward runs the saved-statistics kernel (None where
        # that kernel does not apply: other head sizes, fp32)
        o, lse = K.mhla_attn_fwd(qkv, B, L, H, hd, self.W, mask, pa, sa, want_lse=True) if training else \
            (K.mhla_attn_fwd(qkv, B, L, H, hd, self.W, mask, pa, sa), None)
        wp_c = wcast(wp)
        y = lin_fwd(o, wp_c, bp.detach(), M, D, D, torch.float32, residual=residual, drop=(pp, sp), site=wp)
        return y, (xn, weff, qkv, o, wp_c, mask, B, L, pa, sa, pp, sp, prm, lse)

    @staticmethod
    def out_dropout(saved):
        """(p, seed) of the dropout on this branch's output (the mask its incoming gradient must carry)."""
        return saved[10], saved[11]

    def bwd(self, saved, dy_lp, premasked=False):
        xn, weff, qkv, o, wp_c, mask, B, L, pa, sa, pp, sp, prm, lse = saved
        wqkv, bqkv, wl, bl, wp, bp = prm
        M, D = xn.shape
        H, hd = self.H, D // self.H
        dym = K.dropout(dy_lp, pp, sp) if (pp > 0 and not premasked) else dy_lp
        do = lin_bwd_x(dym, wp_c, M, D, D, xn.dtype, site=wp)
        dwp, dbp = lin_bwd_w(dym, o, M, D, D, wp=wp, bp=bp)
        dqkv = K.mhla_attn_bwd(qkv, do, B, L, H, hd, self.W, mask, pa, sa, o=o if lse is not None else None, lse=lse)
        dxn = lin_bwd_x(dqkv, weff, M, 3 * D, D, xn.dtype, site=wqkv)
        dweff, dbeff = lin_bwd_w(dqkv, xn, M, 3 * D, D)
        # (dW2, dW1, dWproj, dWeff join the grouped weight-gradient launch, which may wait for further blocks; the
        # batched fold below consumes dWeff only after that launch -- EncoderOp.bwd flushes in that order)
        tg = [_gt(p) for p in (wqkv, bqkv, wl, bl)]
        if all(t is not None for t in tg) and _DEFER["on"]:
            _DEFER["fold"].append((dweff, dbeff, wqkv.detach(), bqkv.detach(), wl.detach(), tg, H, (wqkv, bqkv, wl, bl)))
            return dxn, [None, None, None, None, dwp, dbp]
        if (_DEFER["on"] and not wqkv.requires_grad and not bqkv.requires_grad and tg[2] is not None and tg[3] is not None):
            # frozen qkv projection, trainable latent_proj (experiments/sppp_mhla_pretrained.py:236-247): the batched
            # launch with its qkv half switched off, straight into the latent_proj gradient buffers
            _DEFER["fold"].append((dweff, dbeff, wqkv.detach(), bqkv.detach(), wl.detach(), [None, None, tg[2], tg[3]], H,
                                   (wl, bl)))
            return dxn, [None, None, None, None, dwp, dbp]
        flush_wgrads()                      # the fold runs now: dWeff must have been launched
        if all(t is not None for t in tg):
            with _side_stream(dweff, dbeff):
                K.mhla_fold_bwd(dweff, dbeff, wqkv, bqkv, wl, H, out=tg)
                _ready(wqkv, bqkv, wl, bl)
            return dxn, [None, None, None, None, dwp, dbp]
        join_side_stream()
        dwqkv, dbqkv, dwl, dbl = K.mhla_fold_bwd(dweff, dbeff, wqkv, bqkv, wl, H)
        return dxn, [dwqkv, dbqkv, dwl, dbl, dwp, dbp]


class DenseChain:
    q8_fwd, q8_bwd = 0, 2
    """vit.MultiHeadAttention.forward (models/vit.py:77-104) and, with the
    nn.MultiheadAttention parameter names, the use_mhla=False branch
    (models/vit_mhla.py:57-62,96-101).  mask: key-keep [B,L] (torch MHA) or None."""

    def __init__(self, H, p_attn=0.0, p_proj=0.0, torch_mha=False):
        self.H, self.p_attn, self.p_proj, self.torch_mha = H, p_attn, p_proj, torch_mha
        self.names = (("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias") if torch_mha
                      else ("qkv.weight", "qkv.bias", "proj.weight", "proj.bias"))

    def _views(self, t, B, L, D, hd):
        ld = 3 * D
        return (_View(t, 0, ld, L * ld, hd), _View(t, D, ld, L * ld, hd), _View(t, 2 * D, ld, L * ld, hd))

    def fwd(self, xn, prm, B, L, residual, mask, training, ln=None):
        wqkv, bqkv, wp, bp = prm
        M, D = xn.shape if ln is None else ln[0].shape
        H, hd = self.H, D // self.H
        pa = self.p_attn if training else 0.0
        pp = self.p_proj if (training and not self.torch_mha) else 0.0   # nn.MultiheadAttention has no proj dropout
        sa = _seed() if pa > 0 else 0
        sp = _seed() if pp > 0 else 0
        wqkv_c, wp_c = wcast(wqkv), wcast(wp)
        if ln is not None:                       # LayerNorm fused into the qkv projection (see MHLAChain.fwd)
            qkv, _, xn, mu, rs = lin_fwd_ln(ln[:3], wqkv_c, bqkv.detach(), M, 3 * D, D, get_compute_dtype())
            ln[3][:] = [xn, mu, rs]
        else:
            qkv = lin_fwd(xn, wqkv_c, bqkv.detach(), M, 3 * D, D, xn.dtype, site=wqkv)
        o = torch.empty((M, D), dtype=xn.dtype, device=xn.device)
        q, k, v = self._views(qkv, B, L, D, hd)
        ov = _View(o, 0, D, L * D, hd)
        att = sdpa_fwd(q, k, v, ov, B, H, L, L, hd, hd ** -0.5, mask, L if mask is not None else 0, 0, pa, sa)
        y = lin_fwd(o, wp_c, bp.detach(), M, D, D, torch.float32, residual=residual, drop=(pp, sp), site=wp)
        return y, (xn, wqkv_c, qkv, o, wp_c, att, mask, B, L, pa, sa, pp, sp, prm)

    @staticmethod
    def out_dropout(saved):
        return saved[11], saved[12]

    def bwd(self, saved, dy_lp, premasked=False):
        xn, wqkv_c, qkv, o, wp_c, att, mask, B, L, pa, sa, pp, sp, prm = saved
        wqkv, bqkv, wp, bp = prm
        M, D = xn.shape
        H, hd = self.H, D // self.H
        dym = K.dropout(dy_lp, pp, sp) if (pp > 0 and not premasked) else dy_lp
        do = lin_bwd_x(dym, wp_c, M, D, D, xn.dtype, site=wp)
        dwp, dbp = lin_bwd_w(dym, o, M, D, D, wp=wp, bp=bp)
        dqkv = torch.empty_like(qkv)
        q, k, v = self._views(qkv, B, L, D, hd)
        dq, dk, dv = self._views(dqkv, B, L, D, hd)
        sdpa_bwd(q, k, v, _View(o, 0, D, L * D, hd), _View(do, 0, D, L * D, hd), dq, dk, dv, att, B, H, L, L, hd,
                 hd ** -0.5, mask, L if mask is not None else 0, 0, pa, sa)
        dxn = lin_bwd_x(dqkv, wqkv_c, M, 3 * D, D, xn.dtype, site=wqkv)
        dwqkv, dbqkv = lin_bwd_w(dqkv, xn, M, 3 * D, D, wp=wqkv, bp=bqkv)
        return dxn, [dwqkv, dbqkv, dwp, dbp]


class CrossChain:
    """CrossAttention.forward (models/attention.py:37-78; ONE head, scores / embed_dim**0.5,
    no dropout after out_proj) and MultiHeadCrossAttention.forward (attention.py:105-148)."""
    names = ("q_proj.weight", "q_proj.bias", "k_proj.weight", "k_proj.bias", "v_proj.weight", "v_proj.bias",
             "out_proj.weight", "out_proj.bias")

    def __init__(self, H, p_attn=0.0):
        self.H, self.p_attn = H, p_attn

    def fwd(self, qn, kn, prm, B, Lq, Lk, residual, mask, training):
        wq, bq, wk, bk, wv, bv, wo, bo = prm
        D = qn.shape[1]
        H, hd = self.H, D // self.H
        pa = self.p_attn if training else 0.0
        sa = _seed() if pa > 0 else 0
        cdt = qn.dtype
        wq_c, wk_c, wv_c, wo_c = wcast(wq), wcast(wk), wcast(wv), wcast(wo)
        q = lin_fwd(qn, wq_c, bq.detach(), B * Lq, D, D, cdt)
        k = lin_fwd(kn, wk_c, bk.detach(), B * Lk, D, D, cdt)
        v = lin_fwd(kn, wv_c, bv.detach(), B * Lk, D, D, cdt)
        o = torch.empty((B * Lq, D), dtype=cdt, device=qn.device)
        scale = 1.0 / (hd ** 0.5)
        att = sdpa_fwd(_View(q, 0, D, Lq * D, hd), _View(k, 0, D, Lk * D, hd), _View(v, 0, D, Lk * D, hd),
                       _View(o, 0, D, Lq * D, hd), B, H, Lq, Lk, hd, scale, mask,
                       Lq * Lk if mask is not None else 0, Lk if mask is not None else 0, pa, sa)
        y = lin_fwd(o, wo_c, bo.detach(), B * Lq, D, D, torch.float32, residual=residual)
        return y, (qn, kn, q, k, v, o, (wq_c, wk_c, wv_c, wo_c), att, mask, B, Lq, Lk, pa, sa, scale, prm)

    def bwd(self, saved, dy_lp):
        qn, kn, q, k, v, o, (wq_c, wk_c, wv_c, wo_c), att, mask, B, Lq, Lk, pa, sa, scale, prm = saved
        wq, bq, wk, bk, wv, bv, wo, bo = prm
        D = qn.shape[1]
        H, hd = self.H, D // self.H
        cdt = qn.dtype
        Mq, Mk = B * Lq, B * Lk
        do = lin_bwd_x(dy_lp, wo_c, Mq, D, D, cdt)
        dwo, dbo = lin_bwd_w(dy_lp, o, Mq, D, D, wp=wo, bp=bo)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        sdpa_bwd(_View(q, 0, D, Lq * D, hd), _View(k, 0, D, Lk * D, hd), _View(v, 0, D, Lk * D, hd),
                 _View(o, 0, D, Lq * D, hd), _View(do, 0, D, Lq * D, hd), _View(dq, 0, D, Lq * D, hd),
                 _View(dk, 0, D, Lk * D, hd), _View(dv, 0, D, Lk * D, hd), att, B, H, Lq, Lk, hd, scale, mask,
                 Lq * Lk if mask is not None else 0, Lk if mask is not None else 0, pa, sa)
        dqn = lin_bwd_x(dq, wq_c, Mq, D, D, cdt)
        dwq, dbq = lin_bwd_w(dq, qn, Mq, D, D, wp=wq, bp=bq)
        dkn = torch.empty((Mk, D), dtype=torch.float32, device=qn.device)
        K.gemm(dk, wk_c, dkn, Mk, D, D, D, D, D, b_kmajor=False)
        K.gemm(dv, wv_c, dkn, Mk, D, D, D, D, D, b_kmajor=False, accumulate=True)
        dwk, dbk = lin_bwd_w(dk, kn, Mk, D, D, wp=wk, bp=bk)
        dwv, dbv = lin_bwd_w(dv, kn, Mk, D, D, wp=wv, bp=bv)
        return dqn, dkn, [dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo]


class MLPChain:
    q8_fwd, q8_bwd = 0, 2
    """MLP.forward fc1 -> GELU -> drop -> fc2 -> drop (models/vit.py:124-139)."""

    def __init__(self, p=0.0, fc1="fc1", fc2="fc2"):
        self.p = p
        self.names = (f"{fc1}.weight", f"{fc1}.bias", f"{fc2}.weight", f"{fc2}.bias")

    def fwd(self, xn, prm, residual, training, ln=None):
        w1, b1, w2, b2 = prm
        M, D = xn.shape if ln is None else ln[0].shape
        Hd = w1.shape[0]
        Do = w2.shape[0]
        p = self.p if training else 0.0
        s1 = _seed() if p > 0 else 0
        s2 = _seed() if p > 0 else 0
        w1_c, w2_c = wcast(w1), wcast(w2)
        # Low-precision path: the fc1 epilogue saves GELU'(u) (Phi and phi share one exponential) instead of u, so the
        # backward epilogue is a multiply: the GELU arithmetic (~20 VALU slots per element) runs once instead of twice
        # (measured: -0.09 ms per cfg2 step; both epilogues stay bound by their 310 MB of HBM traffic).  The fp32 parity mode keeps the pre-activation and the exact erf.
        if ln is not None:                       # LayerNorm fused into fc1 (bf16 mode only: see EncoderOp.fwd)
            sg = True
            h, pre, xn, mu, rs = lin_fwd_ln(ln[:3], w1_c, b1.detach(), M, Hd, D, get_compute_dtype(), act=ACT_GELU_SAVEGRAD,
                                            want_pre=True, drop=(p, s1))
            ln[3][:] = [xn, mu, rs]
        else:
            sg = xn.dtype != torch.float32
            h, pre = lin_fwd(xn, w1_c, b1.detach(), M, Hd, D, xn.dtype, act=ACT_GELU_SAVEGRAD if sg else ACT_GELU,
                             want_pre=True, drop=(p, s1), site=w1)
        y = lin_fwd(h, w2_c, b2.detach(), M, Do, Hd, torch.float32, residual=residual, drop=(p, s2), site=w2)
        return y, (xn, (pre, sg), h, w1_c, w2_c, p, s1, s2, prm)

    @staticmethod
    def out_dropout(saved):
        return saved[5], saved[7]

    def bwd(self, saved, dy_lp, premasked=False):
        xn, (pre, sg), h, w1_c, w2_c, p, s1, s2, prm = saved
        w1, b1, w2, b2 = prm
        M, D = xn.shape
        Hd, Do = w1_c.shape[0], w2_c.shape[0]
        dym = K.dropout(dy_lp, p, s2) if (p > 0 and not premasked) else dy_lp
        dpre = lin_bwd_x(dym, w2_c, M, Do, Hd, xn.dtype, dgelu_pre=pre, pre_is_grad=sg, drop=(p, s1), site=w2)
        dw2, db2 = lin_bwd_w(dym, h, M, Do, Hd, wp=w2, bp=b2)
        dxn = lin_bwd_x(dpre, w1_c, M, Hd, D, xn.dtype, site=w1)
        dw1, db1 = lin_bwd_w(dpre, xn, M, Hd, D, wp=w1, bp=b1)
        return dxn, [dw1, db1, dw2, db2]


# ------------------------------------------------------------------------------------
# Ops (autograd granularity)
# ------------------------------------------------------------------------------------
class OpFn(torch.autograd.Function):
    """forward(op, n_in, *inputs, *params); the Op owns the kernel sequences."""

    @staticmethod
    def forward(ctx, op, n_in, *tensors):
        K.require_gpu(*[t for t in tensors if t is not None])
        ins, prm = tensors[:n_in], tensors[n_in:]
        out, saved = op.fwd(ins, prm)
        ctx.op, ctx.saved, ctx.n_in, ctx.n_prm = op, saved, n_in, len(prm)
        return out

    @staticmethod
    def backward(ctx, dout):
        din, dprm = ctx.op.bwd(ctx.saved, _as_f32(dout), ctx.needs_input_grad[2:2 + ctx.n_in])
        join_side_stream()
        ctx.saved = None
        return (None, None, *din, *dprm)


def run(op, inputs: Sequence[torch.Tensor], params: Sequence[torch.Tensor]) -> torch.Tensor:
    return OpFn.apply(op, len(inputs), *inputs, *params)


class LinearOp:
    """nn.Linear on fp32 activations (head, standalone projections)."""

    def __init__(self, act=ACT_NONE):
        self.act = act

    def fwd(self, ins, prm):
        (x,), (w, b) = ins, prm
        shp = x.shape
        N = w.shape[0]
        if (self.act == ACT_NONE and N <= K.SMALL_LINEAR_MAX_N and x.dtype == torch.float32 and w.dtype == torch.float32
                and (b is None or b.dtype == torch.float32)):
            # a few classes on a batch of CLS rows: exact-fp32 dot products in one launch instead of one 128x128 GEMM tile
            # behind two casts (favit_small_linear_*: 13-21 us per GEMM launch -> 3-4 us), in every compute mode
            x2 = x.reshape(-1, shp[-1])
            if x2.stride(-1) != 1:
                x2 = x2.contiguous()
            y = K.small_linear_fwd(x2, w.detach(), None if b is None else b.detach())
            return y.reshape(*shp[:-1], N), ("small", x2, shp, b is not None, (w, b))
        a = _as_cdt(x.reshape(-1, shp[-1]))
        w_c = wcast(w)
        M, Kd, N = a.shape[0], a.shape[1], w.shape[0]
        y = lin_fwd(a, w_c, None if b is None else b.detach(), M, N, Kd, torch.float32, allow_fp8=False)
        return y.reshape(*shp[:-1], N), (a, w_c, shp, b is not None, (w, b))

    def bwd(self, saved, dy, needs):
        if isinstance(saved[0], str):
            _, x2, shp, has_b, (w, b) = saved
            M, N = x2.shape[0], w.shape[0]
            dy2 = _as_f32(dy.reshape(M, N))
            if not dy2.is_contiguous():
                dy2 = dy2.contiguous()
            tw, tb = _gt(w), (_gt(b) if has_b else None)
            direct = tw is not None and (tb is not None or not has_b)
            dx, dw, db = K.small_linear_bwd(dy2, x2, w.detach(), want_dx=bool(needs[0]), want_db=has_b,
                                            dw_out=tw if direct else None, db_out=tb if direct else None)
            if direct:
                _ready(w, *( [b] if has_b else [] ))
            dx = dx.reshape(shp) if dx is not None else None
            return [dx], [dw, db] if has_b else [dw]
        a, w_c, shp, has_b, (w, b) = saved
        M, Kd, N = a.shape[0], a.shape[1], w_c.shape[0]
        dy_c = _as_cdt(dy.reshape(M, N))
        dx = lin_bwd_x(dy_c, w_c, M, N, Kd, torch.float32, allow_fp8=False).reshape(shp) if needs[0] else None
        dw, db = lin_bwd_w(dy_c, a, M, N, Kd, want_bias=has_b, wp=w, bp=b, allow_fp8=False)
        return [dx], [dw, db] if has_b else [dw]


class PatchEmbedOp:
    """PatchEmbedding.forward (models/vit.py:43-53): rearrange + Linear(P*P*C -> D)."""

    def __init__(self, P):
        self.P = P

    def fwd(self, ins, prm):
        (img,), (w, b) = ins, prm
        B, Cc, HW, _ = img.shape
        patches = K.patchify_fwd(_as_f32(img), self.P, get_compute_dtype())
        M, Kd = patches.shape
        D = w.shape[0]
        w_c = wcast(w)
        tok = lin_fwd(patches, w_c, b.detach(), M, D, Kd, torch.float32, allow_fp8=False)
        return tok.reshape(B, M // B, D), (patches, w_c, (B, Cc, HW), (w, b))

    def bwd(self, saved, dy, needs):
        patches, w_c, (B, Cc, HW), (w, b) = saved
        M, Kd = patches.shape
        D = w_c.shape[0]
        dy_c = _as_cdt(dy.reshape(M, D))
        dw, db = lin_bwd_w(dy_c, patches, M, D, Kd, wp=w if w.is_leaf else None, bp=b, allow_fp8=False)
        dimg = None
        if needs[0]:
            dpatch = lin_bwd_x(dy_c, w_c, M, D, Kd, torch.float32, allow_fp8=False)
            dimg = K.patchify_bwd(dpatch, B, Cc, HW, self.P)
        return [dimg], [dw, db]


class PrologueOp:
    """cat(cls, tokens) (+ pos_embed) (models/vit.py:292-296, models/sppp_mhla.py:303-304)."""

    def __init__(self, has_pos):
        self.has_pos = has_pos

    def fwd(self, ins, prm):
        (tok,) = ins
        cls = prm[0]
        pos = prm[1] if self.has_pos else None
        B, N, D = tok.shape
        x = K.embed_prologue_fwd(_as_f32(tok), cls.detach().contiguous(), None if pos is None else pos.detach().contiguous(),
                                 B, N, D)
        return x, (B, N, D)

    def bwd(self, saved, dy, needs):
        B, N, D = saved
        dtok, dcls, dpos = K.embed_prologue_bwd(dy, B, N, D, torch.float32, want_pos=self.has_pos)
        out = [dcls.reshape(1, 1, D)]
        if self.has_pos:
            out.append(dpos.reshape(1, N + 1, D))
        return [dtok.reshape(B, N, D)], out


class BlockSpec:
    """Static description of one pre-LN transformer block (models/vit.py:165-179,
    models/vit_mhla.py:77-109, models/mhla.py:205-222)."""

    def __init__(self, attn, mlp):
        self.attn, self.mlp = attn, mlp
        self.names = (("norm1.weight", "norm1.bias") + tuple("attn." + n for n in attn.names) +
                      ("norm2.weight", "norm2.bias") + tuple("mlp." + n for n in mlp.names))
        self.n = len(self.names)


class EncoderOp:
    """A stack of pre-LN blocks on the fp32 residual stream [B, L, D]:
    x += attn(LN1(x)); x += mlp(LN2(x))."""

    def __init__(self, blocks: List[BlockSpec], mask=None, training=False):
        self.blocks, self.mask, self.training = blocks, mask, training

    def fwd(self, ins, prm):
        (x,) = ins
        B, L, D = x.shape
        M = B * L
        cdt = get_compute_dtype()
        x = _as_f32(x).reshape(M, D)
        tapes = []
        # short token counts, bf16: the two LayerNorms of a block ride in the A-operand staging of the qkv / fc1
        # projections (favit_ln_gemm; the library declines shapes beyond the 64-row kernel's regime, and
        # lin_fwd_ln then issues the two launches)
        fuse = _LN_FUSE and cdt == torch.bfloat16 and get_compute_mode() == "bf16" and M <= 16384 and D % 64 == 0 and D <= 512
        # the latent_proj folds only depend on parameters: all MHLA blocks of equal geometry in ONE launch
        pre = [None] * len(self.blocks)
        idx, fp, off = [], [], 0
        for bi, bs in enumerate(self.blocks):
            if isinstance(bs.attn, MHLAChain):
                idx.append(bi)
                fp.append(tuple(prm[off + 2:off + 6]))
            off += bs.n
        if 2 <= len(idx) <= 32 and len({(self.blocks[i].attn.H, tuple(q[0].shape)) for i, q in zip(idx, fp)}) == 1:
            for bi, we in zip(idx, K.mhla_fold_fwd_multi(fp, self.blocks[idx[0]].attn.H, cdt)):
                pre[bi] = we
        off = 0
        for bi, bs in enumerate(self.blocks):
            p = list(prm[off:off + bs.n])
            off += bs.n
            na = len(bs.attn.names)
            g1, b1 = p[0], p[1]
            pa = p[2:2 + na]
            g2, b2 = p[2 + na], p[3 + na]
            pm = p[4 + na:]
            if fuse and isinstance(bs.attn, (MHLAChain, DenseChain)):
                o1 = [None, None, None]
                kw = {"pre": pre[bi]} if pre[bi] is not None else {}
                x1, sa = bs.attn.fwd(None, pa, B, L, x, self.mask, self.training, ln=(x, g1.detach(), b1.detach(), o1), **kw)
                xn1, mu1, rs1 = o1
            else:
                xn1, mu1, rs1 = K.layernorm_fwd(x, D, g1, b1, M, D, cdt, q8=_q8_request(bs.attn, pa, "fwd", D))
                if pre[bi] is not None:
                    x1, sa = bs.attn.fwd(xn1, pa, B, L, x, self.mask, self.training, pre=pre[bi])
                else:
                    x1, sa = bs.attn.fwd(xn1, pa, B, L, x, self.mask, self.training)
            if fuse and isinstance(bs.mlp, MLPChain):
                o2 = [None, None, None]
                x2, sm = bs.mlp.fwd(None, pm, x1, self.training, ln=(x1, g2.detach(), b2.detach(), o2))
                xn2, mu2, rs2 = o2
            else:
                xn2, mu2, rs2 = K.layernorm_fwd(x1, D, g2, b2, M, D, cdt, q8=_q8_request(bs.mlp, pm, "fwd", D))
                x2, sm = bs.mlp.fwd(xn2, pm, x1, self.training)
            tapes.append((x, mu1, rs1, (g1, b1), sa, x1, mu2, rs2, (g2, b2), sm, (pa, pm)))
            x = x2
        return x.reshape(B, L, D), (tapes, B, L, D)

    def bwd(self, saved, dy, needs):
        tapes, B, L, D = saved
        M = B * L
        g = dy.reshape(M, D)
        g_lp = _as_cdt(g)
        grads = []
        begin_wgrads()
        begin_deferred()
        begin_zero_pool(sum(3 * D + 4 for bs in self.blocks if isinstance(bs.attn, MHLAChain)), dy.device)

        def ln_bwd(dxn, xin, gam, bet, mu, rs, dres, pd, q8=None):
            """LayerNorm backward of the stream; the dgamma / dbeta fold joins the deferred batch when the parameters own
            gradient buffers (the fused-optimizer flow).  q8: the low-precision copy also leaves quantised (fp8 mode)."""
            if not gam.requires_grad and not bet.requires_grad:      # frozen layer: no dgamma / dbeta fold at all
                return K.layernorm_bwd(dxn, xin, D, gam, mu, rs, M, D, dres=dres, want_lp=True, lp_drop=pd, frozen=True, q8=q8)
            tg_, tb_ = _gt(gam), _gt(bet)
            lst = _DEFER["ln"] if (_DEFER["on"] and tg_ is not None and tb_ is not None) else None
            n0 = len(lst) if lst is not None else 0
            out = K.layernorm_bwd(dxn, xin, D, gam, mu, rs, M, D, dres=dres, want_lp=True, dg_out=tg_, db_out=tb_,
                                  lp_drop=pd, defer=lst, q8=q8)
            if lst is not None:
                lst[n0] = lst[n0] + ((gam, bet),)
            elif out[2] is None:
                _ready(gam, bet)
            return out
        try:
            # The low-precision copy of the stream gradient only feeds the next branch's backward GEMMs; when that
            # branch's output was dropped in forward, the LayerNorm backward that produces the copy applies the mask
            # (no separate masking pass: 24 launches of 28 us per cfg2 step at dropout 0.1).
            order = list(zip(reversed(self.blocks), reversed(tapes)))
            premasked = False
            for bi, (bs, tp) in enumerate(order):
                x, mu1, rs1, (g1, b1), sa, x1, mu2, rs2, (g2, b2), sm, (pa_, _) = tp
                dxn2, gm = bs.mlp.bwd(sm, g_lp, premasked=premasked)
                pd = bs.attn.out_dropout(sa) if hasattr(bs.attn, "out_dropout") else (0.0, 0)
                # (fp8 mode: the copy feeds this block's proj input-gradient GEMM as its dY -- consumed as it is
                # written, since a branch with output dropout gets it pre-masked)
                g, g_lp, dg2, db2 = ln_bwd(dxn2, x1, g2, b2, mu2, rs2, g, pd,
                                           q8=_q8_request(bs.attn, pa_, "bwd", D) if hasattr(bs.attn, "out_dropout") else None)
                dxn1, ga = (bs.attn.bwd(sa, g_lp, premasked=pd[0] > 0) if hasattr(bs.attn, "out_dropout")
                            else bs.attn.bwd(sa, g_lp))
                pd = (0.0, 0)
                q8n = None
                if bi + 1 < len(order):
                    nbs, ntp = order[bi + 1]
                    pd = nbs.mlp.out_dropout(ntp[9])
                    q8n = _q8_request(nbs.mlp, ntp[10][1], "bwd", D)       # the next block's fc2 input-gradient GEMM
                premasked = pd[0] > 0
                g, g_lp, dg1, db1 = ln_bwd(dxn1, x, g1, b1, mu1, rs1, g, pd, q8=q8n)
                if flush_wgrads(force=False) and _STATE["grad_ready"] is not None:
                    flush_deferred()
                if not _SIDE["enabled"]:
                    join_side_stream()
                grads = [dg1, db1] + ga + [dg2, db2] + gm + grads
        except BaseException:
            _WG["list"] = None                # drop the half-collected weight gradients of the failed backward
            _SIDE["pending"].clear()
            _DEFER.update(on=False, fold=[], ln=[])
            end_zero_pool()
            raise
        end_wgrads()
        end_deferred()
        end_zero_pool()
        join_side_stream()
        return [g.reshape(B, L, D)], grads


class FinalNormOp:
    """norm(x)[:, 0] (models/vit.py:303-307): only the CLS row of the last LayerNorm is computed."""

    def fwd(self, ins, prm):
        (x,), (g, b) = ins, prm
        B, L, D = x.shape
        x = _as_f32(x)
        y, mu, rs = K.layernorm_fwd(x, L * D, g, b, B, D, torch.float32)
        return y, (x, mu, rs, (g, b), (B, L, D))

    def bwd(self, saved, dy, needs):
        x, mu, rs, (g, b), (B, L, D) = saved
        dx = torch.zeros((B, L, D), dtype=torch.float32, device=x.device)
        _, _, dg, db = K.layernorm_bwd(dy, x, L * D, g, mu, rs, B, D, dx=dx, lddx=L * D, dg_out=_gt(g), db_out=_gt(b))
        if dg is None:
            _ready(g, b)
        return [dx], [dg, db]


class AttnOp:
    """Stand-alone attention module on fp32 [B, L, D] (no LayerNorm, no residual)."""

    def __init__(self, chain, mask=None, training=False):
        self.chain, self.mask, self.training = chain, mask, training

    def fwd(self, ins, prm):
        (x,) = ins
        B, L, D = x.shape
        xn = _as_cdt(x.reshape(B * L, D))
        y, saved = self.chain.fwd(xn, list(prm), B, L, None, self.mask, self.training)
        return y.reshape(B, L, D), (saved, (B, L, D))

    def bwd(self, saved, dy, needs):
        s, (B, L, D) = saved
        dxn, grads = self.chain.bwd(s, _as_cdt(dy.reshape(B * L, D)))
        return [_as_f32(dxn).reshape(B, L, D)], grads


class MLPOp:
    def __init__(self, chain, training=False):
        self.chain, self.training = chain, training

    def fwd(self, ins, prm):
        (x,) = ins
        shp = x.shape
        xn = _as_cdt(x.reshape(-1, shp[-1]))
        y, saved = self.chain.fwd(xn, list(prm), None, self.training)
        return y.reshape(*shp[:-1], y.shape[-1]), (saved, shp)

    def bwd(self, saved, dy, needs):
        s, shp = saved
        dxn, grads = self.chain.bwd(s, _as_cdt(dy.reshape(-1, dy.shape[-1])))
        return [_as_f32(dxn).reshape(shp)], grads


class CrossAttnOp:
    """Stand-alone (MultiHead)CrossAttention on fp32 query [B,Lq,D], key_value [B,Lk,D]."""

    def __init__(self, chain, mask=None, training=False):
        self.chain, self.mask, self.training = chain, mask, training

    def fwd(self, ins, prm):
        q, kv = ins
        B, Lq, D = q.shape
        Lk = kv.shape[1]
        qn, kn = _as_cdt(q.reshape(B * Lq, D)), _as_cdt(kv.reshape(B * Lk, D))
        y, saved = self.chain.fwd(qn, kn, list(prm), B, Lq, Lk, None, self.mask, self.training)
        return y.reshape(B, Lq, D), (saved, (B, Lq, Lk, D))

    def bwd(self, saved, dy, needs):
        s, (B, Lq, Lk, D) = saved
        dqn, dkn, grads = self.chain.bwd(s, _as_cdt(dy.reshape(B * Lq, D)))
        return [_as_f32(dqn).reshape(B, Lq, D), dkn.reshape(B, Lk, D)], grads


class CrossBlockOp:
    """CrossAttentionTransformerBlock.forward (models/attention.py:194-219)."""

    def __init__(self, chain, mlp, mask=None, training=False):
        self.chain, self.mlp, self.mask, self.training = chain, mlp, mask, training
        self.names = (("norm1_query.weight", "norm1_query.bias", "norm1_kv.weight", "norm1_kv.bias") +
                      tuple("attn." + n for n in chain.names) + ("norm2.weight", "norm2.bias") +
                      tuple("mlp." + n for n in mlp.names))

    def fwd(self, ins, prm):
        q, kv = ins
        B, Lq, D = q.shape
        Lk = kv.shape[1]
        Mq, Mk = B * Lq, B * Lk
        cdt = get_compute_dtype()
        p = list(prm)
        q2, kv2 = _as_f32(q).reshape(Mq, D), _as_f32(kv).reshape(Mk, D)
        qn, muq, rsq = K.layernorm_fwd(q2, D, p[0], p[1], Mq, D, cdt)
        kn, muk, rsk = K.layernorm_fwd(kv2, D, p[2], p[3], Mk, D, cdt)
        x1, sa = self.chain.fwd(qn, kn, p[4:12], B, Lq, Lk, q2, self.mask, self.training)
        xn2, mu2, rs2 = K.layernorm_fwd(x1, D, p[12], p[13], Mq, D, cdt)
        x2, sm = self.mlp.fwd(xn2, p[14:], x1, self.training)
        return x2.reshape(B, Lq, D), (q2, kv2, muq, rsq, muk, rsk, sa, x1, mu2, rs2, sm, p, (B, Lq, Lk, D))

    def bwd(self, saved, dy, needs):
        q2, kv2, muq, rsq, muk, rsk, sa, x1, mu2, rs2, sm, p, (B, Lq, Lk, D) = saved
        Mq, Mk = B * Lq, B * Lk
        g = dy.reshape(Mq, D)
        dxn2, gm = self.mlp.bwd(sm, _as_cdt(g))
        g, g_lp, dg2, db2 = K.layernorm_bwd(dxn2, x1, D, p[12], mu2, rs2, Mq, D, dres=g, want_lp=True)
        dqn, dkn, ga = self.chain.bwd(sa, g_lp)
        dq, _, dgq, dbq = K.layernorm_bwd(dqn, q2, D, p[0], muq, rsq, Mq, D, dres=g)
        dkv, _, dgk, dbk = K.layernorm_bwd(dkn, kv2, D, p[2], muk, rsk, Mk, D)
        return [dq.reshape(B, Lq, D), dkv.reshape(B, Lk, D)], [dgq, dbq, dgk, dbk] + ga + [dg2, db2] + gm


class PoolOp:
    """SuperpixelPooling.pool over a batch (models/sppp.py:192-223, models/sppp_mhla.py:286-300)."""

    def __init__(self, kind, perm, offs, R):
        self.kind, self.perm, self.offs, self.R = kind, perm, offs, R

    def fwd(self, ins, prm):
        (emb,) = ins
        emb = _as_f32(emb)
        out, argmax = K.sppp_pool_fwd(emb, self.perm, self.offs, self.kind, self.R)
        return out, (emb, argmax)

    def bwd(self, saved, dy, needs):
        emb, argmax = saved
        return [K.sppp_pool_bwd(dy, emb, self.perm, self.offs, argmax, self.kind, self.R)], []


class PosEncOp:
    """DynamicPositionalEncoding.forward, centroid branch (models/sppp.py:267-300)."""

    def __init__(self, cent):
        self.cent = cent

    def fwd(self, ins, prm):
        (x,) = ins
        return K.sppp_posenc_fwd(_as_f32(x), self.cent), None

    def bwd(self, saved, dy, needs):
        return [dy], []


class DropoutOp:
    """nn.Dropout on an fp32 activation (embedding dropout, models/vit.py:297)."""

    def __init__(self, p):
        self.p = p

    def fwd(self, ins, prm):
        s = _seed()
        return K.dropout(_as_f32(ins[0]), self.p, s), s

    def bwd(self, saved, dy, needs):
        return [K.dropout(dy, self.p, saved)], []
