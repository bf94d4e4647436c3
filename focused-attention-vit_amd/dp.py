"""Data-parallel gradient synchronisation: one process per GPU, ``torch.distributed`` with the
``nccl`` backend (= RCCL over xGMI on MI355X); ``gloo`` on CPU in the tests.

The reference has no distributed code at all; the only exchange the path needs is
the gradient all-reduce after ``loss.backward()`` (every image is independent: no BatchNorm,
no cross-sample op).  Design for xGMI (7 point-to-point links per GPU, no switch):
  * gradients live in ONE flat fp32 buffer per parameter group (``.grad`` tensors are views),
    so a bucket is a slice -- no gather/scatter copies around the collective;
  * few, large buckets: on a fully connected 8-GPU node RCCL's all-reduce is per-link bound, so large
    messages amortise the per-collective launch + latency -- but the LAST bucket (patch embedding + first
    block) cannot overlap with anything, so the default is an eighth of the gradient bytes, clamped to
    4..32 MiB (ViT-MHLA-Small: 88 MB of gradients -> 8 buckets of ~11 MiB; ViT-Base: 32 MiB);
  * a bucket's all-reduce is issued (async, on RCCL's own stream) as soon as every parameter in it
    has reported a gradient (a set of parameters, not a count of events: a parameter used twice in
    one forward reports twice), which overlaps it with the rest of backward.  Gradient accumulation
    over several backward passes must wrap all but the last in ``no_sync()``; a gradient that arrives
    after its bucket was launched raises instead of producing rank-divergent sums.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import os
import threading

import torch
import torch.distributed as dist

# FAVIT_DP_DEBUG=<dir>: every rank appends its sequence of bucket launches / waits to <dir>/dp_rank<r>.log
# (collectives must be issued in the same order on every rank; this is how a mismatch is found)
_DEBUG = os.environ.get("FAVIT_DP_DEBUG")
# FAVIT_DP_VERIFY=1: ordering check of the overlapped path.  When a bucket's all-reduce is launched, a snapshot of the
# slice is enqueued on the compute stream (i.e. after every kernel that wrote it, by stream order); finish() then
# reduces the snapshots synchronously, after a full device sync, and compares with what the asynchronous collectives
# delivered.  A collective that read its slice before the last writer had finished, or a writer that ran after the
# launch, shows up as a mismatch (it would otherwise be a silently wrong gradient on RCCL).  Costs one extra copy and
# one extra all-reduce per bucket: a test / bring-up switch, never on in a timed run.
_VERIFY = bool(os.environ.get("FAVIT_DP_VERIFY"))
# FAVIT_DP_NO_DRAIN=1: gloo only, diagnostic -- do not drain the compute stream before a host-staged collective
_NO_DRAIN = bool(os.environ.get("FAVIT_DP_NO_DRAIN"))


def _dbg(msg):
    r = dist.get_rank() if dist.is_initialized() else 0
    with open(os.path.join(_DEBUG, f"dp_rank{r}.log"), "a") as f:
        f.write(msg + "\n")


def buf_is_cuda(t) -> bool:
    return bool(t.is_cuda)


class FlatBuffers:
    """Re-homes parameters (and their gradients) into contiguous flat fp32 buffers.

    Parameters stay ordinary ``nn.Parameter`` objects with ordinary shapes (callers keep
    ``.data.copy_``/``state_dict`` semantics); only their storage becomes a view of ``flat_p``.
    Order inside the buffer = reverse registration order ~ the order gradients become ready.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.params.reverse()
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, torch.float32
        n = 0
        self.offsets = []
        for p in self.params:
            if p.dtype != dt:
                raise TypeError("parameters must be fp32 masters")
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.numel = n
        self.flat_p = torch.zeros(n, dtype=dt, device=dev)
        self.flat_g = torch.zeros(n, dtype=dt, device=dev)
        for p, o in zip(self.params, self.offsets):
            v = self.flat_p[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    def zero_grad(self):
        self.flat_g.zero_()
        for p, o in zip(self.params, self.offsets):      # re-attach if a caller dropped .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)


class _Staged:
    """Host-staged (gloo) collectives of device buffers, process-wide FIFO.  An entry is (sync, bucket, slice, event):
    the event was recorded on the compute stream behind the slice's last writer.  pump() issues, in order, every
    entry whose event has completed -- from an otherwise idle stream, so that gloo's own "wait for the caller's
    stream" is satisfied the moment it is recorded and NO device-side wait is ever queued (see GradSync._launch for
    what those waits did to a 4-rank rehearsal).  It is called from the thread that runs backward whenever a gradient
    is reported, i.e. every few kernels; finish() pumps with block=True (host wait on the events).  One queue for all
    parameter groups: the order of collectives is the order of the _launch calls, identical on every rank.  (A helper
    thread doing the host waits was tried first: correct, but every hand-over of the interpreter lock to it costs the
    5 ms switch interval while the main thread is busy launching kernels -- 179 ms per step instead of 66.)"""
    pending = None
    side = None

    @classmethod
    def push(cls, item):
        if cls.pending is None:
            import collections
            cls.pending = collections.deque()
        cls.pending.append(item)

    @classmethod
    def pump(cls, block=False):
        q = cls.pending
        while q:
            sync, b, buf, ev = q[0]
            if block:
                ev.synchronize()
            elif not ev.query():
                return
            q.popleft()
            if cls.side is None or cls.side.device != buf.device:
                cls.side = torch.cuda.Stream(device=buf.device)
            with torch.cuda.stream(cls.side):
                h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=sync.group, async_op=True)
            sync._handles.append((h, buf))
            if _DEBUG:
                _dbg(f"issued sync#{id(sync) % 9973} bucket {b}")


class GradSync:
    """Bucketed, overlapped all-reduce (mean) of a FlatBuffers' gradient buffer."""

    def __init__(self, flat: FlatBuffers, bucket_mb: Optional[float] = None,
                 group: Optional[dist.ProcessGroup] = None, wire_dtype: Optional[torch.dtype] = None):
        """wire_dtype=torch.bfloat16 (or FAVIT_DP_WIRE=bf16): the buckets travel as bf16 -- half the bytes per xGMI
        link, which is what bounds a ring all-reduce on this node (at 17 tokens per image the 88 MB
        fp32 exchange is as long as the step it follows).  The kernels keep accumulating into the fp32 flat buffer; at
        launch a bucket is rounded into a persistent bf16 wire buffer (stream-ordered behind its last writer), the
        collective sums bf16, and finish() widens the reduced values back into the fp32 buffer the fused AdamW reads
        (moments and master weights stay fp32).  Deviation from the fp32 exchange: one bf16 rounding of every rank's
        contribution plus the collective's bf16 partial sums, <= ~world x 2^-9 relative per element
        (tests/test_dp_gloo.py bounds it at world_size 2).  Opt-in: the default exchange is fp32."""
        self.flat, self.group = flat, group
        if wire_dtype is None and os.environ.get("FAVIT_DP_WIRE", "").lower() in ("bf16", "bfloat16"):
            wire_dtype = torch.bfloat16
        if wire_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("GradSync wire_dtype must be None / torch.float32 / torch.bfloat16")
        self.wire_dtype = None if wire_dtype == torch.float32 else wire_dtype
        self._wire = torch.zeros(flat.numel, dtype=self.wire_dtype, device=flat.flat_g.device) if self.wire_dtype is not None else None
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # FAVIT_DP_FORCE=1: issue the collectives in a one-rank group too (tests/test_dp_gpu.py: the real RCCL call
        # path -- async all-reduce of flat-buffer slices from the autograd thread -- on a one-GPU box)
        self._active = self.world > 1 or (dist.is_initialized() and bool(os.environ.get("FAVIT_DP_FORCE")))
        self._host_staged = dist.is_initialized() and dist.get_backend(group) == "gloo"
        if bucket_mb is None:
            bucket_mb = min(32.0, max(4.0, flat.numel * 4 / (1 << 20) / 8))
        self.bucket_mb = bucket_mb
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        # bucket boundaries on parameter boundaries
        self.buckets = []          # (start, end, [param indices])
        start, idxs = 0, []
        for i, (p, o) in enumerate(zip(flat.params, flat.offsets)):
            idxs.append(i)
            end = o + (p.numel() + 3) // 4 * 4
            if end - start >= cap:
                self.buckets.append((start, end, idxs))
                start, idxs = end, []
        if idxs:
            self.buckets.append((start, flat.numel, idxs))
        self._bucket_of = {}
        for b, (_, _, idxs) in enumerate(self.buckets):
            for i in idxs:
                self._bucket_of[i] = b
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._launched = [False] * len(self.buckets)
        self._hooks = []
        self.defer = bool(os.environ.get("FAVIT_DP_DEFER"))     # debugging aid: launch every bucket from finish()
        self._index = {id(p): i for i, p in enumerate(flat.params)}
        if self._active:
            for i, p in enumerate(flat.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def grad_ready(self, p):
        """Called by the kernels' direct gradient-accumulation path (functional.set_grad_ready_hook)
        for parameters autograd never sees a gradient for."""
        i = self._index.get(id(p))
        if i is not None and self._active:
            self._event(i, p, True)

    def no_sync(self):
        """Context manager for gradient accumulation: backward passes inside it only accumulate into the
        flat buffer; no bucket is launched.  The LAST micro-batch's backward runs outside it (its
        ready events launch the buckets, overlapped with that backward), or call finish() directly."""
        sync = self

        class _NoSync:
            def __enter__(self):
                self.prev, sync.defer = sync.defer, True

            def __exit__(self, *exc):
                sync.defer = self.prev
                return False
        return _NoSync()

    def _make_hook(self, i):
        def hook(_p):
            self._event(i, _p, False)
        return hook

    def _event(self, i, p, direct):
        """Parameter i has (another contribution to) its gradient in the flat buffer.  direct: reported by a kernel
        launch sequence that wrote the gradient itself; otherwise autograd's post-accumulate hook."""
        if self.defer:                 # graph capture / replay / no_sync(): finish() launches every bucket afterwards
            return
        if self._host_staged and _Staged.pending:
            _Staged.pump()             # staged buckets whose writers have finished meanwhile go out now
        b = self._bucket_of[i]
        prev = self._seen[b].get(i)
        if prev is not None:
            if prev and not direct:
                # autograd runs the AccumulateGrad node (and this hook) of a parameter even when the custom Function
                # returned None for it -- i.e. right after the kernels reported the same gradient directly
                return
            # A real second contribution: a second backward before step() (gradient accumulation without no_sync())
            # or a parameter used twice in one forward.  If the bucket is already on the wire the reduced values
            # would miss it and differ across ranks.
            if self._launched[b]:
                raise RuntimeError(
                    f"GradSync: gradient of parameter #{i} {tuple(p.shape)} accumulated after its bucket's all-reduce "
                    "was launched.  Either a second backward of this step ran outside no_sync() -- wrap all but the "
                    "last backward of a step in GradSync.no_sync() (FusedAdamW.no_sync()) -- or the parameter is used "
                    "more than once in one forward (weight sharing), in which case its bucket must not go out on the "
                    "first contribution: set GradSync.defer = True (every bucket is then launched from finish())")
            return
        self._seen[b][i] = direct
        if len(self._seen[b]) == self._pending[b]:
            self._launch(b)

    def reset(self):
        for b, (_, _, idxs) in enumerate(self.buckets):
            self._pending[b] = len(idxs)          # distinct parameters whose gradient must have arrived
            self._launched[b] = False
        self._seen = [dict() for _ in self.buckets]      # parameter index -> reported directly by the kernels?
        self._handles = []
        self._snap = []
        self._widen = []                                  # (fp32 slice, bf16 wire slice) of every bucket launched in bf16

    def _launch(self, b):
        if self._launched[b] or not self._active:
            return
        s, e, _ = self.buckets[b]
        buf = self.flat.flat_g[s:e]
        self._launched[b] = True
        if _DEBUG:
            _dbg(f"launch sync#{id(self) % 9973} bucket {b} [{s}:{e}) thread {threading.current_thread().name}")
        if _VERIFY:
            self._snap.append((b, buf.clone()))
        if self._wire is not None:
            # bf16 on the wire: round the bucket into the wire buffer on the compute stream (behind its last writer by
            # stream order); the collective and everything below work on that slice, finish() widens it back
            wire = self._wire[s:e]
            wire.copy_(buf)
            self._widen.append((buf, wire))
            buf = wire
        if self._host_staged and buf.is_cuda and not _NO_DRAIN:
            # gloo stages device tensors through pinned host memory: the op makes one of gloo's pool streams WAIT (on
            # the device) for an event recorded on the calling thread's current stream -- here the compute stream, with
            # the rest of backward queued behind it -- and a gloo worker thread blocks in a stream synchronise until the
            # copy has run.  With several ranks sharing ONE GPU (the only way to rehearse N ranks on a one-GPU box)
            # those device-side waits sit at the head of hardware queues of four processes at once and the box stops
            # making progress: a 4-rank run without this block hung in its third step with, on EVERY rank, the main
            # thread in wait() on the first bucket's handle, two threads in kfd_wait_on_events and two spinning on HSA
            # signals (gloo's workers inside their stream synchronise), the network thread idle in epoll
            # (thread dumps of bench.py's FAVIT_BENCH_WATCHDOG).  So no device-side wait is created: an event
            # is recorded behind the bucket's last writer and the collective is issued only once that event has
            # completed (polled from this thread at every reported gradient: _Staged), from an otherwise idle stream
            # (gloo's own event is complete the moment it is recorded).  Buckets still go out while backward runs.  RCCL (one process per GPU, collectives ordered by
            # stream events on its own stream) keeps the direct path below.
            ev = torch.cuda.Event()
            ev.record()
            _Staged.push((self, b, buf, ev))
            _Staged.pump()
            return
        self._handles.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), buf))

    def _drain_helper(self):
        """Every staged bucket of the process has been issued."""
        _Staged.pump(block=True)

    def abort(self):
        """A backward that raised may leave asynchronous all-reduces outstanding on slices of flat_g: wait for them
        (the buffer is about to be zeroed or reused) and forget the step's bookkeeping."""
        try:
            self._drain_helper()
        except Exception:          # noqa: BLE001
            pass
        for h, _ in self._handles:
            try:
                h.wait()
            except Exception:      # noqa: BLE001  (a failed collective must not mask the original error)
                pass
        self.reset()

    def _verify(self):
        """FAVIT_DP_VERIFY: the asynchronously reduced slices against a synchronous reduction of the snapshots."""
        if buf_is_cuda(self.flat.flat_g):
            torch.cuda.synchronize()
        worst = 0.0
        for b, snap in self._snap:
            s, e, _ = self.buckets[b]
            ref = snap.clone()
            dist.all_reduce(ref, op=dist.ReduceOp.SUM, group=self.group)
            got = self.flat.flat_g[s:e]
            err = (got - ref).abs().max().item()
            scale = ref.abs().max().item()
            worst = max(worst, err / max(scale, 1e-30))
            tol = 1e-5 if self._wire is None else 2.0 ** -7 * max(2, self.world)      # bf16 wire: rounding, not ordering
            if err > tol * max(scale, 1e-30):
                raise RuntimeError(f"FAVIT_DP_VERIFY: bucket {b} [{s}:{e}) differs from the synchronous reduction of its "
                                   f"launch-time snapshot by {err:.3e} (max |ref| {scale:.3e}): a writer of this slice was "
                                   "not ordered before the collective")
        self.verified = getattr(self, "verified", 0) + len(self._snap)
        if _DEBUG:
            _dbg(f"verify sync#{id(self) % 9973}: {len(self._snap)} buckets, worst relative difference {worst:.2e}")

    def finish(self, average: bool = True):
        """Wait for every bucket (launching the ones whose hooks did not fire, e.g. frozen or
        unused parameters).  average=True turns the sums into means in place; the fused
        optimizer passes average=False and folds 1/world into its gradient scale instead."""
        if self._active:
            for b in range(len(self.buckets)):
                self._launch(b)
            self._drain_helper()
            for k, (h, _) in enumerate(self._handles):
                if _DEBUG:
                    _dbg(f"wait sync#{id(self) % 9973} handle {k}/{len(self._handles)}")
                h.wait()
            if _DEBUG:
                _dbg(f"done sync#{id(self) % 9973}")
            for dst, wire in self._widen:                 # the reduced bf16 values back into the fp32 gradient buffer
                dst.copy_(wire)
            if _VERIFY:
                self._verify()
            if average:
                self.flat.flat_g.mul_(1.0 / self.world)
        self.reset()
