"""Tensor-level wrappers over the C ABI (include/favit.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
function below hands raw device pointers to libfavit.so.  All tensors must live on a
ROCm device; there is no CPU path (the oracle in ``oracle/`` is test infrastructure and
is never imported from the product).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _abi
from ._abi import ACT_DGELU, ACT_GELU, ACT_NONE, BF16, F32, GemmDesc, SdpaDesc

_DT = {torch.float32: F32, torch.bfloat16: BF16}

# bench.py sets this to a list to time every favit_gemm launch with HIP events recorded on the
# stream the kernel is launched on (torch's current stream): entries are
# (start_event, end_event, algorithmic_flops, kernel_key, shape, kernel family the library dispatched to).
GEMM_TRACE = None


def dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if torch.is_tensor(t_or_dtype) else t_or_dtype
    try:
        return _DT[d]
    except KeyError:
        raise TypeError(f"favit kernels support float32 / bfloat16 only, got {d}") from None


def require_gpu(*ts):
    """Every tensor must live on the CURRENT ROCm device: kernels are launched on the current device's
    current stream (_st), so a tensor of another device would be dereferenced on the wrong GPU."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "focused-attention-vit_amd runs its hot path in HIP kernels on an MI355X; got a CPU tensor. "
                "There is no CPU fallback (move the module and its inputs to the GPU).")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError(
                f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: one process drives one GPU "
                "(torch.cuda.set_device(local_rank) before building the model), or wrap the call in "
                "`with torch.cuda.device(t.device):`")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# --------------------------------------------------------------------------------------
def gemm(A, B, Cc, M, N, K, lda, ldb, ldc, *, a_kmajor=True, b_kmajor=True, bias=None, act=ACT_NONE,
         aux_in=None, ld_aux_in=0, aux_out=None, ld_aux_out=0, residual=None, ld_res=0, a_rowsum=None,
         accumulate=False, alpha=1.0, batch=1, batch_inner=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), split_k=0,
         a_off=0, b_off=0, c_off=0, dropout_p=0.0, dropout_seed=0, scale_a=None, scale_b=None):
    """C[m,n] = epilogue(alpha * sum_k A[m,k] B[n,k]).  Offsets/strides are in elements.
    float8 operands (A e4m3 / e5m2, B e4m3; both k-major): scale_a / scale_b are the device scalars
    written by fp8_quantize."""
    require_gpu(A, B, Cc)
    fp8 = A.dtype in _FP8_DT
    if fp8:
        if B.dtype != torch.float8_e4m3fn:
            raise TypeError("fp8 gemm: B must be float8_e4m3fn")
    elif A.dtype != B.dtype:
        raise TypeError("gemm operands must share a dtype")
    d = GemmDesc()
    ea, ec = A.element_size(), Cc.element_size()
    d.A = A.data_ptr() + a_off * ea
    d.B = B.data_ptr() + b_off * ea
    d.C = Cc.data_ptr() + c_off * ec
    d.bias = bias.data_ptr() if bias is not None else None
    d.aux_in = aux_in.data_ptr() if aux_in is not None else None
    d.aux_out = aux_out.data_ptr() if aux_out is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.a_rowsum = a_rowsum.data_ptr() if a_rowsum is not None else None
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.ld_aux_in, d.ld_aux_out, d.ld_res = ld_aux_in, ld_aux_out, ld_res
    d.sAo, d.sAi = sA
    d.sBo, d.sBi = sB
    d.sCo, d.sCi = sC
    d.batch, d.batch_inner = batch, batch_inner
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.in_dtype, d.out_dtype = (_abi.FP8 if fp8 else dt(A)), dt(Cc)
    if fp8:
        d.fp8_fmt = 1 if A.dtype == torch.float8_e5m2 else 0
        d.scale_a = scale_a.data_ptr() if scale_a is not None else None
        d.scale_b = scale_b.data_ptr() if scale_b is not None else None
    d.act = act
    d.accumulate = int(accumulate)
    d.split_k = split_k
    d.alpha = alpha
    d.dropout_p = dropout_p
    d.dropout_seed = dropout_seed
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("bias must be fp32")
    if residual is not None and residual.dtype != torch.float32:
        raise TypeError("residual must be fp32")
    if GEMM_TRACE is None:
        _abi.check(_abi.lib().favit_gemm(C.byref(d), _st()), "favit_gemm")
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _abi.check(_abi.lib().favit_gemm(C.byref(d), _st()), "favit_gemm")
    e1.record()
    key = ("fp8" if fp8 else "bf16" if A.dtype == torch.bfloat16 else "f32") + ("_K" if a_kmajor else "_M") + ("K" if b_kmajor else "M") + \
          ("_obf16" if Cc.dtype == torch.bfloat16 else "_of32")
    GEMM_TRACE.append((e0, e1, 2.0 * M * N * K * batch, key, (M, N, K, batch), _abi.lib().favit_gemm_last_kernel().decode()))


def ln_gemm(x, ldx, gamma, beta, w, out, M, N, D, *, bias=None, act=ACT_NONE, aux_out=None, residual=None,
            dropout_p=0.0, dropout_seed=0, eps=1e-5):
    """out = epilogue(LayerNorm(x) @ w^T) in one launch (favit_ln_gemm); returns (xn bf16 [M, D], mean, rstd), or None
    when the library declines the shape (the caller then runs layernorm_fwd + gemm)."""
    require_gpu(x, gamma, beta, w, out)
    if w.dtype != torch.bfloat16 or x.dtype != torch.float32:
        return None
    d = GemmDesc()
    d.A = None
    d.B, d.C = w.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.aux_out = aux_out.data_ptr() if aux_out is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.M, d.N, d.K = M, N, D
    d.lda, d.ldb, d.ldc = D, w.stride(0), out.stride(0)
    d.ld_aux_out, d.ld_res = N, N
    d.batch, d.batch_inner = 1, 1
    d.a_kmajor, d.b_kmajor = 1, 1
    d.in_dtype, d.out_dtype = BF16, dt(out)
    d.act, d.alpha = act, 1.0
    d.dropout_p, d.dropout_seed = dropout_p, dropout_seed
    xn = torch.empty((M, D), dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    if GEMM_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _abi.lib().favit_ln_gemm(C.byref(d), _p(x), ldx, _p(gamma), _p(beta), eps, _p(xn), _p(mean), _p(rstd), _st())
    if rc == -2:
        return None
    _abi.check(rc, "favit_ln_gemm")
    if GEMM_TRACE is not None:
        e1.record()
        GEMM_TRACE.append((e0, e1, 2.0 * M * N * D, "bf16_KK_" + ("obf16" if out.dtype == torch.bfloat16 else "of32"),
                           (M, N, D, 1), "s64ln"))
    return xn, mean, rstd


_GROUPED_WS = {}          # device index -> workspace tensor of the slab-mode split-K reduction
# Workspaces that were outgrown stay alive: a captured HIP graph holds their ADDRESS (train.GraphedStep: one graph per
# token-count bucket, the second bucket can need a larger workspace than the first), and torch.cuda.graph() empties the
# allocator cache when the next capture begins -- a freed workspace is then unmapped under the first graph's replays
# (memory access fault; found with the two-bucket cfg5 bench once its one-problem launches took the slab path).
_GROUPED_WS_RETIRED = []


GROUP_MAX = 48            # problems per grouped launch (csrc/gemm.hip)


def gemm_grouped_tn(problems, use_workspace: bool = True) -> bool:
    """One launch for several weight-gradient GEMMs dW = dY^T X that share the token dim (more than GROUP_MAX problems:
    consecutive launches of GROUP_MAX).
    problems: list of (dy [T,N], a [T,K], dw [N,K] fp32, db [N] fp32 or None, accumulate: bool).
    Returns False if the library cannot group them (caller falls back to single launches).
    use_workspace: when the library splits the token dimension, the partial results of the K-splits go through a
    cached device workspace and are summed in a fixed order (deterministic, no fp32 atomics); False keeps the atomic
    path.  With enough tiles in the launch there is ONE split and the tiles write dW themselves (no workspace)."""
    if len(problems) > GROUP_MAX:
        T = problems[0][0].shape[0]
        if any(p[0].shape[0] != T or p[0].dtype != torch.bfloat16 or p[1].dtype != torch.bfloat16 for p in problems) or T % 32:
            return False                               # decline as a whole: never half a list
        for i in range(0, len(problems), GROUP_MAX):
            if not gemm_grouped_tn(problems[i:i + GROUP_MAX], use_workspace):
                raise RuntimeError("grouped weight-gradient launch declined a chunk after launching another")
        return True
    n = len(problems)
    if n == 0:
        return True
    arr = (GemmDesc * n)()
    T = problems[0][0].shape[0]
    for d, (dy, a, dw, db, acc) in zip(arr, problems):
        require_gpu(dy, a, dw)
        if dy.dtype != torch.bfloat16 or a.dtype != torch.bfloat16 or dw.dtype != torch.float32 or dy.shape[0] != T:
            return False
        N, Kd = dy.shape[1], a.shape[1]
        d.A, d.B, d.C = dy.data_ptr(), a.data_ptr(), dw.data_ptr()
        d.a_rowsum = db.data_ptr() if db is not None else None
        d.M, d.N, d.K = N, Kd, T
        d.lda, d.ldb, d.ldc = dy.stride(0), a.stride(0), dw.stride(0)
        d.batch, d.batch_inner = 1, 1
        d.a_kmajor, d.b_kmajor = 0, 0
        d.in_dtype, d.out_dtype = BF16, F32
        d.act, d.accumulate, d.split_k = ACT_NONE, int(acc), 0
        d.alpha, d.dropout_p, d.dropout_seed = 1.0, 0.0, 0
    if GEMM_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    need = int(_abi.lib().favit_gemm_grouped_tn_workspace(arr, n)) if use_workspace else 0
    if need > 0:
        dev = problems[0][0].device
        ws = _GROUPED_WS.get(dev.index)
        if ws is None or ws.numel() < need:
            if ws is not None:
                _GROUPED_WS_RETIRED.append(ws)
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            _GROUPED_WS[dev.index] = ws
        rc = _abi.lib().favit_gemm_grouped_tn_ws(arr, n, _p(ws), ws.numel(), _st())
    else:
        rc = _abi.lib().favit_gemm_grouped_tn(arr, n, _st())
    if rc == -2:
        return False
    _abi.check(rc, "favit_gemm_grouped_tn")
    if GEMM_TRACE is not None:
        e1.record()
        fl = sum(2.0 * p[0].shape[0] * p[0].shape[1] * p[1].shape[1] for p in problems)
        GEMM_TRACE.append((e0, e1, fl, "bf16_MM_of32_grouped", (len(problems), int(_abi.lib().favit_gemm_grouped_last_splits())),
                           "grouped_tn"))
    return True


_FP8_DT = (torch.float8_e4m3fn, torch.float8_e5m2)


class Fp8History:
    """Delayed per-tensor scaling state of one quantisation site: three rotating amax slots (this call's scale
    comes from the amax the previous call measured; this call measures into the next slot and clears the one
    after) and the scale_inv scalar the GEMM reads."""
    __slots__ = ("slots", "scale_inv", "calls")

    NSLOT = 256                                   # FAVIT_FP8_AMAX_SLOTS: partial maxima per array

    def __init__(self, device):
        self.slots = torch.zeros(3 * self.NSLOT, dtype=torch.float32, device=device)
        self.scale_inv = torch.ones(1, dtype=torch.float32, device=device)
        self.calls = 0


def fp8_quantize(src: torch.Tensor, fmt: torch.dtype, *, want=True, want_t=False, colsum: Optional[torch.Tensor] = None,
                 hist: Optional[Fp8History] = None):
    """Per-tensor scaled conversion of a [rows, cols] fp32 / bf16 matrix to OCP fp8 (`fmt` =
    torch.float8_e4m3fn | torch.float8_e5m2).  Returns (q [rows, cols] or None, q_t [cols, ld_t] or None
    with ld_t = rows rounded up to 64 and the pad zero-filled, scale_inv device scalar).  Without `hist` the amax is
    taken on the device in the same call sequence (a separate pass, no host sync).  With `hist` (delayed scaling) the
    scale comes from the amax the site's previous call measured and this call measures the next one while it
    quantises: one pass over the tensor; values beyond the previous amax saturate.  colsum [cols] fp32 (optional) gets
    the column sums of src ADDED (bias gradient)."""
    require_gpu(src)
    if src.dim() != 2 or src.stride(1) != 1:
        raise ValueError("fp8_quantize expects a row-major 2-D matrix")
    rows, cols = src.shape
    dev = src.device
    L = _abi.lib()
    q = torch.empty((rows, cols), dtype=fmt, device=dev) if want else None
    ld_t = (rows + 63) // 64 * 64
    q_t = torch.empty((cols, ld_t), dtype=fmt, device=dev) if want_t else None
    f8 = _abi.E5M2 if fmt == torch.float8_e5m2 else _abi.E4M3
    if hist is None:
        st = torch.zeros(2, dtype=torch.float32, device=dev)        # [amax, scale_inv]
        _abi.check(L.favit_fp8_amax(_p(src), dt(src), rows, cols, src.stride(0), _p(st), _st()), "favit_fp8_amax")
        _abi.check(L.favit_fp8_quantize(_p(src), dt(src), rows, cols, src.stride(0), _p(q), cols, _p(q_t), ld_t, f8,
                                        _p(st), C.c_void_p(st.data_ptr() + 4), _p(colsum), None, None, _st()),
                   "favit_fp8_quantize")
        return q, q_t, st[1:2]
    base = hist.slots.data_ptr()
    stride = 4 * hist.NSLOT
    cur, nxt, clr = hist.calls % 3, (hist.calls + 1) % 3, (hist.calls + 2) % 3
    if hist.calls == 0:                                             # no history yet: measure this tensor first
        _abi.check(L.favit_fp8_amax(_p(src), dt(src), rows, cols, src.stride(0), C.c_void_p(base + stride * cur), _st()),
                   "favit_fp8_amax")
    hist.calls += 1
    sinv = torch.empty(1, dtype=torch.float32, device=dev)          # per call: the GEMM reads it after later calls
    _abi.check(L.favit_fp8_quantize(_p(src), dt(src), rows, cols, src.stride(0), _p(q), cols, _p(q_t), ld_t, f8,
                                    C.c_void_p(base + stride * cur), _p(sinv), _p(colsum), C.c_void_p(base + stride * nxt),
                                    C.c_void_p(base + stride * clr), _st()), "favit_fp8_quantize")
    return q, q_t, sinv


def cast(src: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    require_gpu(src)
    src = src.contiguous()
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    _abi.check(_abi.lib().favit_cast(_p(src), dt(src), _p(out), dt(out), src.numel(), _st()), "favit_cast")
    return out


def _q8_slots(hist: "Fp8History"):
    """Rotate a delayed-scaling history by one call: (amax, amax_next, amax_clear) pointers of THIS call."""
    base, stride = hist.slots.data_ptr(), 4 * hist.NSLOT
    cur, nxt, clr = hist.calls % 3, (hist.calls + 1) % 3, (hist.calls + 2) % 3
    hist.calls += 1
    return C.c_void_p(base + stride * cur), C.c_void_p(base + stride * nxt), C.c_void_p(base + stride * clr)


def _q8_fused_ok(q8, out_dtype, D) -> bool:
    """A LayerNorm pass may quantise its bf16 output for the consumer site: the site has a measured amax (its first
    call takes the two-pass form, which measures before it scales) and the rows are whole 4-byte groups."""
    return q8 is not None and q8[1] is not None and q8[1].calls > 0 and out_dtype == torch.bfloat16 and D % 4 == 0


def layernorm_fwd(x, ldx, gamma, beta, rows, D, out_dtype, eps=1e-5, q8=None):
    """x: fp32 rows with stride ldx -> y [rows, D] (out_dtype), mean, rstd.
    q8 = (fp8 dtype, Fp8History of the consumer GEMM's operand site) in fp8 mode: y also leaves the pass quantised
    (favit_layernorm_fwd_q8) and the result is parked on the tensor where functional._q8 finds it."""
    require_gpu(x, gamma, beta)
    y = torch.empty((rows, D), dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    if _q8_fused_ok(q8, out_dtype, D):
        fmt, hist = q8
        q = torch.empty((rows, D), dtype=fmt, device=x.device)
        sinv = torch.empty(1, dtype=torch.float32, device=x.device)
        cur, nxt, clr = _q8_slots(hist)
        _abi.check(_abi.lib().favit_layernorm_fwd_q8(_p(x), ldx, _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, D, eps,
                                                     _p(q), _abi.E5M2 if fmt == torch.float8_e5m2 else _abi.E4M3, cur,
                                                     _p(sinv), nxt, clr, _st()), "favit_layernorm_fwd_q8")
        y._favit_q8 = ((fmt, y._version), (q, None, sinv))
        return y, mean, rstd
    _abi.check(_abi.lib().favit_layernorm_fwd(_p(x), ldx, _p(gamma), _p(beta), _p(y), dt(y), _p(mean), _p(rstd),
                                              rows, D, eps, _st()), "favit_layernorm_fwd")
    return y, mean, rstd


SMALL_LINEAR_MAX_N = 64


def small_linear_fwd(x, w, bias):
    """y = x @ w.T + bias in exact fp32 (x [M, K] fp32 contiguous rows, w [N, K] fp32, N <= 64): the classification head."""
    require_gpu(x, w)
    M, Kd = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _abi.check(_abi.lib().favit_small_linear_fwd(_p(x), x.stride(0), _p(w), _p(bias), _p(y), M, N, Kd, _st()),
               "favit_small_linear_fwd")
    return y


def small_linear_bwd(dy, x, w, *, want_dx=True, dw_out=None, db_out=None, want_db=True):
    """(dx or None, dw or None, db or None) of small_linear_fwd.  dw_out / db_out: fp32 gradient buffers the results are
    ACCUMULATED into (then None is returned in their place)."""
    require_gpu(dy, x, w)
    M, Kd = x.shape
    N = w.shape[0]
    dev = x.device
    dx = torch.empty((M, Kd), dtype=torch.float32, device=dev) if want_dx else None
    acc = dw_out is not None and (db_out is not None or not want_db)
    dw = dw_out if acc else torch.empty((N, Kd), dtype=torch.float32, device=dev)
    db = (db_out if acc else torch.empty(N, dtype=torch.float32, device=dev)) if want_db else None
    _abi.check(_abi.lib().favit_small_linear_bwd(_p(dy), _p(x), x.stride(0), _p(w), _p(dx), _p(dw), _p(db), int(acc), M, N, Kd,
                                                 _st()), "favit_small_linear_bwd")
    return dx, (None if acc else dw), (None if (acc or not want_db) else db)


def layernorm_bwd(dy, x, ldx, gamma, mean, rstd, rows, D, *, dres=None, dx=None, lddx=None, want_lp=False,
                  dg_out=None, db_out=None, lp_drop=(0.0, 0), defer=None, frozen=False, q8=None):
    """Returns dx (fp32, row stride lddx), dx_lp (dy.dtype copy or None), dgamma, dbeta.
    dg_out / db_out: optional fp32 [D] gradient buffers the affine gradients are ACCUMULATED into
    (then None is returned in their place).  lp_drop = (p, seed): dx_lp carries that dropout mask (the branch it
    feeds had its output dropped in forward), saving a separate masking pass.
    defer (a list, with dg_out / db_out): the fold of the partial sums into dg_out / db_out is NOT launched; the
    entry (part, dg_out, db_out) is appended and the caller folds several of them in one launch (reduce_rows_multi).
    frozen: gamma and beta take no gradient (fine-tuning with frozen layers): no fold launch at all, (.., None, None).
    q8 = (fp8 dtype, Fp8History) in fp8 mode: dx_lp also leaves the pass quantised (see layernorm_fwd)."""
    require_gpu(dy, x)
    dev = x.device
    if dx is None:
        dx = torch.empty((rows, D), dtype=torch.float32, device=dev)
        lddx = D
    dx_lp = torch.empty((rows, D), dtype=dy.dtype, device=dev) if want_lp else None
    nparts = int(min(2048, (rows + 3) // 4))
    part = torch.empty((2, nparts, D), dtype=torch.float32, device=dev)
    acc = dg_out is not None and db_out is not None
    if frozen:
        acc, dg, db = True, None, None
    elif acc and defer is not None:
        defer.append((part, dg_out, db_out))
        dg = db = None
    elif acc:
        dg, db = dg_out, db_out
    else:
        dgb = torch.empty((2, D), dtype=torch.float32, device=dev)
        dg, db = dgb[0], dgb[1]
    if want_lp and _q8_fused_ok(q8, dy.dtype, D):
        fmt, hist = q8
        q = torch.empty((rows, D), dtype=fmt, device=dev)
        sinv = torch.empty(1, dtype=torch.float32, device=dev)
        cur, nxt, clr = _q8_slots(hist)
        _abi.check(_abi.lib().favit_layernorm_bwd_q8(_p(dy), _p(x), ldx, _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), lddx,
                                                     _p(dx_lp), _p(part[0]), _p(part[1]), nparts, _p(dg), _p(db), int(acc), rows, D,
                                                     float(lp_drop[0]), int(lp_drop[1]) & ((1 << 64) - 1), _p(q),
                                                     _abi.E5M2 if fmt == torch.float8_e5m2 else _abi.E4M3, cur, _p(sinv), nxt, clr,
                                                     _st()), "favit_layernorm_bwd_q8")
        dx_lp._favit_q8 = ((fmt, dx_lp._version), (q, None, sinv))
        return (dx, dx_lp, None, None) if acc else (dx, dx_lp, dg, db)
    _abi.check(_abi.lib().favit_layernorm_bwd(_p(dy), dt(dy), _p(x), ldx, _p(gamma), _p(mean), _p(rstd), _p(dres),
                                              _p(dx), lddx, _p(dx_lp), dt(dy), _p(part[0]), _p(part[1]), nparts,
                                              _p(dg), _p(db), int(acc), rows, D, float(lp_drop[0]), int(lp_drop[1]) & ((1 << 64) - 1),
                                              _st()), "favit_layernorm_bwd")
    return (dx, dx_lp, None, None) if acc else (dx, dx_lp, dg, db)


def reduce_rows_multi(entries):
    """entries: [(part [2, rows, cols] fp32 contiguous, out0 [cols], out1 [cols])] -> out0 += colsum(part[0]),
    out1 += colsum(part[1]) for every entry in ONE launch (at most 32 entries per launch)."""
    for i in range(0, len(entries), 32):
        chunk = entries[i:i + 32]
        n = len(chunk)
        rows, cols = chunk[0][0].shape[1], chunk[0][0].shape[2]
        arr = C.c_void_p * n
        for part, o0, o1 in chunk:
            require_gpu(part, o0, o1)
            if tuple(part.shape) != (2, rows, cols) or not part.is_contiguous() or part.dtype != torch.float32:
                raise ValueError("reduce_rows_multi: every entry must be a contiguous fp32 [2, rows, cols] stack of one shape")
        _abi.check(_abi.lib().favit_reduce_rows_multi(n, arr(*[e[0].data_ptr() for e in chunk]),
                                                      arr(*[e[1].data_ptr() for e in chunk]),
                                                      arr(*[e[2].data_ptr() for e in chunk]), rows, cols, _st()),
                   "favit_reduce_rows_multi")


def reduce_rows(t2d: torch.Tensor) -> torch.Tensor:
    require_gpu(t2d)
    rows, cols = t2d.shape
    out = torch.empty(cols, dtype=torch.float32, device=t2d.device)
    _abi.check(_abi.lib().favit_reduce_rows(_p(t2d), t2d.stride(0), _p(out), rows, cols, 0, _st()), "favit_reduce_rows")
    return out


def mhla_fold_fwd(wqkv, bqkv, wl, bl, H, dtype):
    require_gpu(wqkv, bqkv, wl, bl)
    D = wqkv.shape[1]
    weff = torch.empty((3 * D, D), dtype=dtype, device=wqkv.device)
    beff = torch.empty(3 * D, dtype=torch.float32, device=wqkv.device)
    _abi.check(_abi.lib().favit_mhla_fold_fwd(_p(wqkv), _p(bqkv), _p(wl), _p(bl), _p(weff), dt(weff), None, _p(beff),
                                              D, H, _st()), "favit_mhla_fold_fwd")
    return weff, beff


def mhla_fold_fwd_multi(params, H, dtype):
    """params: list of (wqkv, bqkv, wl, bl) per block -> list of (weff, beff); one launch for all blocks."""
    n = len(params)
    D = params[0][0].shape[1]
    dev = params[0][0].device
    for q in params:
        require_gpu(*q)
    weff = torch.empty((n, 3 * D, D), dtype=dtype, device=dev)
    beff = torch.empty((n, 3 * D), dtype=torch.float32, device=dev)
    arr = C.c_void_p * n
    cols = [arr(*[q[j].data_ptr() for q in params]) for j in range(4)]
    wp = arr(*[weff[i].data_ptr() for i in range(n)])
    bp = arr(*[beff[i].data_ptr() for i in range(n)])
    _abi.check(_abi.lib().favit_mhla_fold_fwd_multi(n, cols[0], cols[1], cols[2], cols[3], wp, dt(weff), bp, D, H, _st()),
               "favit_mhla_fold_fwd_multi")
    return [(weff[i], beff[i]) for i in range(n)]


def mhla_fold_bwd(dweff, dbeff, wqkv, bqkv, wl, H, out=None):
    """out = (dwqkv, dbqkv, dwl, dbl) gradient buffers to ACCUMULATE into, or None for fresh tensors."""
    D = wqkv.shape[1]
    hd = D // H
    dev = wqkv.device
    acc = out is not None
    if acc:
        dwqkv, dbqkv, dwl, dbl = out
    else:
        dwqkv = torch.empty((3 * D, D), dtype=torch.float32, device=dev)
        dbqkv = torch.empty(3 * D, dtype=torch.float32, device=dev)
        dwl = torch.empty((hd, hd), dtype=torch.float32, device=dev)
        dbl = torch.empty(hd, dtype=torch.float32, device=dev)
    _abi.check(_abi.lib().favit_mhla_fold_bwd(_p(dweff), _p(dbeff), _p(wqkv), _p(bqkv), _p(wl), _p(dwqkv), _p(dbqkv),
                                              _p(dwl), _p(dbl), D, H, int(acc), _st()), "favit_mhla_fold_bwd")
    return dwqkv, dbqkv, dwl, dbl


def mhla_fold_bwd_multi(entries, H):
    """entries: [(dweff, dbeff, wqkv, bqkv, wl, (dwqkv, dbqkv, dwl, dbl))] of equal geometry: every layer's fold
    backward in ONE launch (chunks of 16), ACCUMULATING into the gradient buffers."""
    for i in range(0, len(entries), 16):
        chunk = entries[i:i + 16]
        n = len(chunk)
        D = chunk[0][2].shape[1]
        arr = C.c_void_p * n
        cols = []
        for j in range(5):
            cols.append(arr(*[e[j].data_ptr() for e in chunk]))
        for j in range(4):     # (dwqkv, dbqkv may be None for every layer: frozen qkv projection, latent_proj gradients only)
            cols.append(arr(*[(e[5][j].data_ptr() if e[5][j] is not None else None) for e in chunk]))
        for e in chunk:
            require_gpu(*e[:5], *e[5])
        _abi.check(_abi.lib().favit_mhla_fold_bwd_multi(n, *cols, D, H, _st()), "favit_mhla_fold_bwd_multi")


def mhla_attn_lse_supported(L, hd, W, dtype) -> bool:
    """Do favit_mhla_attn_fwd_lse / _bwd_lse take this shape (bf16, hd = 64, odd W <= 7 or <= 11 with L > 16)?"""
    code = {torch.float32: _abi.F32, torch.bfloat16: _abi.BF16}.get(dtype)
    return code is not None and bool(_abi.lib().favit_mhla_attn_lse_supported(L, hd, W, code))


def mhla_attn_fwd(qkv, B, L, H, hd, W, mask=None, p=0.0, seed=0, want_lse=False):
    """Attention core; with want_lse returns (out, lse fp32 [B, H, L]) -- lse is None where the saved-statistics
    backward does not apply (the caller then uses mhla_attn_bwd without it)."""
    require_gpu(qkv, mask)
    out = torch.empty((B * L, H * hd), dtype=qkv.dtype, device=qkv.device)
    if want_lse and mhla_attn_lse_supported(L, hd, W, qkv.dtype):
        lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
        _abi.check(_abi.lib().favit_mhla_attn_fwd_lse(_p(qkv), _p(out), _p(lse), _p(mask), B, L, H, hd, W, dt(qkv), p,
                                                      seed, _st()), "favit_mhla_attn_fwd_lse")
        return out, lse
    _abi.check(_abi.lib().favit_mhla_attn_fwd(_p(qkv), _p(out), _p(mask), B, L, H, hd, W, dt(qkv), p, seed, _st()),
               "favit_mhla_attn_fwd")
    return (out, None) if want_lse else out


def mhla_attn_bwd(qkv, dout, B, L, H, hd, W, mask=None, p=0.0, seed=0, o=None, lse=None):
    """dqkv of the attention core; with the forward's output and lse the saved-statistics kernel runs."""
    require_gpu(qkv, dout, mask, o, lse)
    dqkv = torch.empty_like(qkv)
    if o is not None and lse is not None:
        _abi.check(_abi.lib().favit_mhla_attn_bwd_lse(_p(qkv), _p(dout), _p(o), _p(lse), _p(dqkv), _p(mask), B, L, H, hd,
                                                      W, dt(qkv), p, seed, _st()), "favit_mhla_attn_bwd_lse")
        return dqkv
    _abi.check(_abi.lib().favit_mhla_attn_bwd(_p(qkv), _p(dout), _p(dqkv), _p(mask), B, L, H, hd, W, dt(qkv), p, seed,
                                              _st()), "favit_mhla_attn_bwd")
    return dqkv


def _sdpa_desc(q, k, v, o, lse, B, H, Lq, Lk, hd, scale, mask, m_sb, m_sq, p, seed):
    """q, k, v, o: (tensor, element offset, row stride, batch stride, head stride) views (functional._View)."""
    d = SdpaDesc()
    es = q.t.element_size()
    for name, vw in (("q", q), ("k", k), ("v", v), ("o", o)):
        setattr(d, name, vw.t.data_ptr() + vw.off * es)
        getattr(d, name + "_str")[:] = [vw.ld, vw.sb, vw.sh]
    d.lse = lse.data_ptr()
    d.mask = mask.data_ptr() if mask is not None else None
    d.m_sb, d.m_sq = m_sb, m_sq
    d.B, d.H, d.Lq, d.Lk, d.hd, d.dtype = B, H, Lq, Lk, hd, dt(q.t)
    d.scale, d.dropout_p, d.seed = scale, p, seed
    return d


def sdpa_fwd(q, k, v, o, B, H, Lq, Lk, hd, scale, mask=None, m_sb=0, m_sq=0, p=0.0, seed=0):
    """Fused attention forward: writes o (view) and returns lse [B*H, Lq] fp32 (saved for backward)."""
    require_gpu(q.t, k.t, v.t, o.t, mask)
    lse = torch.empty((B * H, Lq), dtype=torch.float32, device=q.t.device)
    d = _sdpa_desc(q, k, v, o, lse, B, H, Lq, Lk, hd, scale, mask, m_sb, m_sq, p, seed)
    _abi.check(_abi.lib().favit_sdpa_fwd(C.byref(d), _st()), "favit_sdpa_fwd")
    return lse


def sdpa_bwd(q, k, v, o, do, dq, dk, dv, lse, B, H, Lq, Lk, hd, scale, mask=None, m_sb=0, m_sq=0, p=0.0, seed=0):
    """Fused attention backward: writes dq, dk, dv (views); probabilities are recomputed from lse."""
    require_gpu(q.t, k.t, v.t, o.t, do.t, dq.t, dk.t, dv.t, lse, mask)
    delta = torch.empty((B * H, Lq), dtype=torch.float32, device=q.t.device)
    d = _sdpa_desc(q, k, v, o, lse, B, H, Lq, Lk, hd, scale, mask, m_sb, m_sq, p, seed)
    es = q.t.element_size()
    for name, vw in (("dout", do), ("dq", dq), ("dk", dk), ("dv", dv)):
        setattr(d, name, vw.t.data_ptr() + vw.off * es)
        getattr(d, ("do" if name == "dout" else name) + "_str")[:] = [vw.ld, vw.sb, vw.sh]
    d.delta = delta.data_ptr()
    _abi.check(_abi.lib().favit_sdpa_bwd(C.byref(d), _st()), "favit_sdpa_bwd")


def softmax_fwd(S, p_dtype, H, Z, Lq, Lk, mask=None, m_sb=0, m_sq=0, p=0.0, seed=0):
    require_gpu(S, mask)
    P = torch.empty((Z, Lq, Lk), dtype=p_dtype, device=S.device)
    Pd = torch.empty_like(P) if p > 0 else None
    _abi.check(_abi.lib().favit_softmax_fwd(_p(S), _p(P), _p(Pd), dt(P), _p(mask), m_sb, m_sq, H, Z, Lq, Lk, p, seed,
                                            _st()), "favit_softmax_fwd")
    return P, (Pd if Pd is not None else P)


def softmax_bwd(P, dPd, Z, Lq, Lk, p=0.0, seed=0):
    dS = torch.empty_like(P)
    _abi.check(_abi.lib().favit_softmax_bwd(_p(P), dt(P), _p(dPd), _p(dS), dt(dS), Z, Lq, Lk, p, seed, _st()),
               "favit_softmax_bwd")
    return dS


def patchify_fwd(img, P, dtype):
    require_gpu(img)
    B, Cc, HW, HW2 = img.shape
    if HW != HW2 or HW % P:
        raise ValueError("patchify needs square images with size divisible by patch_size")
    img = img.contiguous()
    g = HW // P
    out = torch.empty((B * g * g, P * P * Cc), dtype=dtype, device=img.device)
    _abi.check(_abi.lib().favit_patchify_fwd(_p(img), _p(out), dt(out), B, Cc, HW, P, _st()), "favit_patchify_fwd")
    return out


def patchify_bwd(dpatch, B, Cc, HW, P):
    dimg = torch.empty((B, Cc, HW, HW), dtype=torch.float32, device=dpatch.device)
    _abi.check(_abi.lib().favit_patchify_bwd(_p(dpatch), _p(dimg), B, Cc, HW, P, _st()), "favit_patchify_bwd")
    return dimg


def embed_prologue_fwd(tok, cls, pos, B, N, D):
    x = torch.empty((B, N + 1, D), dtype=torch.float32, device=tok.device)
    _abi.check(_abi.lib().favit_embed_prologue_fwd(_p(tok), _p(cls), _p(pos), _p(x), B, N, D, _st()),
               "favit_embed_prologue_fwd")
    return x


def embed_prologue_bwd(dx, B, N, D, tok_dtype, want_pos=True):
    dev = dx.device
    dtok = torch.empty((B * N, D), dtype=tok_dtype, device=dev)
    dcls = torch.empty(D, dtype=torch.float32, device=dev)
    dpos = torch.empty((N + 1, D), dtype=torch.float32, device=dev) if want_pos else None
    _abi.check(_abi.lib().favit_embed_prologue_bwd(_p(dx), _p(dtok), dt(dtok), _p(dcls), _p(dpos), B, N, D, _st()),
               "favit_embed_prologue_bwd")
    return dtok, dcls, dpos


def dropout(x, p, seed):
    y = torch.empty_like(x)
    _abi.check(_abi.lib().favit_dropout(_p(x), _p(y), dt(x), x.numel(), p, seed, _st()), "favit_dropout")
    return y


def sppp_map_patches(seg, P):
    require_gpu(seg)
    if seg.dtype != torch.int64:
        seg = seg.to(torch.int64)
    seg = seg.contiguous()
    B, HW, _ = seg.shape
    N = (HW // P) ** 2
    dev = seg.device
    rank = torch.empty((B, N), dtype=torch.int32, device=dev)
    ntok = torch.empty(B, dtype=torch.int32, device=dev)
    perm = torch.empty((B, N), dtype=torch.int32, device=dev)
    offs = torch.empty((B, N + 1), dtype=torch.int32, device=dev)
    dom = torch.empty((B, N), dtype=torch.int64, device=dev)
    _abi.check(_abi.lib().favit_sppp_map_patches(_p(seg), _p(rank), _p(ntok), _p(perm), _p(offs), _p(dom), B, HW, P,
                                                 _st()), "favit_sppp_map_patches")
    return rank, ntok, perm, offs, dom


def sppp_pool_fwd(emb, perm, offs, kind, R):
    B, N, D = emb.shape
    out = torch.empty((B, R, D), dtype=torch.float32, device=emb.device)
    argmax = torch.empty((B, R, D), dtype=torch.int32, device=emb.device) if kind == 1 else None
    _abi.check(_abi.lib().favit_sppp_pool_fwd(_p(emb), _p(perm), _p(offs), _p(out), _p(argmax), kind, B, N, R, D,
                                              _st()), "favit_sppp_pool_fwd")
    return out, argmax


def sppp_pool_bwd(dout, emb, perm, offs, argmax, kind, R):
    B, N, D = emb.shape
    demb = torch.zeros_like(emb)
    _abi.check(_abi.lib().favit_sppp_pool_bwd(_p(dout), _p(emb), None, _p(perm), _p(offs), _p(argmax), _p(demb), kind,
                                              B, N, R, D, _st()), "favit_sppp_pool_bwd")
    return demb


def sppp_centroids(seg, S):
    require_gpu(seg)
    seg = seg.to(torch.int64).contiguous()
    B, HW, _ = seg.shape
    cent = torch.empty((B, S, 2), dtype=torch.float32, device=seg.device)
    _abi.check(_abi.lib().favit_sppp_centroids(_p(seg), _p(cent), B, HW, S, _st()), "favit_sppp_centroids")
    return cent


def sppp_posenc_fwd(x, cent):
    B, L, D = x.shape
    y = torch.empty_like(x)
    _abi.check(_abi.lib().favit_sppp_posenc_fwd(_p(x), _p(cent), _p(y), B, L, D, 0 if cent is None else cent.shape[1], _st()),
               "favit_sppp_posenc_fwd")
    return y


def slic_grid(H, W, n_segments):
    """Seed grid of skimage.segmentation.slic for a 2-D image (skimage.util.regular_grid on (1, H, W)): returns
    (ys, xs, step) -- seeds at start + i*stride per axis, start = floor(s/2), stride = round(s), s = sqrt(H*W/n)."""
    import math
    if H * W <= n_segments:
        return list(range(H)), list(range(W)), 1
    s = math.sqrt(H * W / float(n_segments))
    sy = sx = s
    if min(H, W) < s:                      # regular_grid: a dimension shorter than the step gets step = its length
        if H <= W:
            sy, sx = float(H), (W / float(n_segments))
        else:
            sx, sy = float(W), (H / float(n_segments))
    ys = list(range(int(sy // 2), H, max(1, int(round(sy)))))
    xs = list(range(int(sx // 2), W, max(1, int(round(sx)))))
    return ys, xs, max(max(1, int(round(sy))), max(1, int(round(sx))))


def slic(images, n_segments=16, compactness=0.1, sigma=1.0, max_num_iter=10, min_size_factor=0.5, stages=False,
         rescale=True):
    """SLIC label maps [B,H,W] int64 of fp32 images [B,3,H,W] on the device (csrc/slic.hip; parity with
    scikit-image unpinned, see include/favit.h).  stages=True also returns (feat, cluster_labels, n_regions).
    rescale (default, scikit-image >= 0.19): every image is first rescaled to [0, 1] by its own min / max over all
    channels, so mean/std-normalised inputs (what the reference's models pass) segment like their [0, 1] originals;
    rescale=False restates scikit-image < 0.19, which takes the values as sRGB in [0, 1] as they are."""
    require_gpu(images)
    if images.dim() != 4 or images.shape[1] != 3 or images.dtype != torch.float32:
        raise TypeError("slic expects fp32 images [B, 3, H, W]")
    images = images.contiguous()
    B, _, H, W = images.shape
    ys, xs, step = slic_grid(H, W, n_segments)
    if len(ys) * len(xs) > 64:
        raise ValueError("slic: at most 64 seed centres are supported")
    dev = images.device
    init = torch.tensor([[y, x] for y in ys for x in xs], dtype=torch.int32, device=dev)
    Kc = init.shape[0]
    coef = int(round((step / float(compactness)) ** 2))
    min_size = int(min_size_factor * (H * W / float(Kc)))
    feat = torch.empty((B, H * W, 4), dtype=torch.int16, device=dev)
    lab = torch.empty((B, H * W), dtype=torch.uint8, device=dev)
    ws = torch.empty((2, B, H * W), dtype=torch.int32, device=dev)
    out = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    nreg = torch.empty(B, dtype=torch.int32, device=dev)
    L = _abi.lib()
    fws = torch.empty(int(L.favit_slic_features_workspace(B, H, W)) // 4, dtype=torch.float32, device=dev)
    _abi.check(L.favit_slic_features(_p(images), _p(feat), B, H, W, float(sigma), int(bool(rescale)), _p(fws), _st()),
               "favit_slic_features")
    cws = torch.empty(int(L.favit_slic_cluster_workspace(Kc, B)) // 8 + 1, dtype=torch.int64, device=dev)
    _abi.check(L.favit_slic_cluster(_p(feat), _p(lab), _p(init), Kc, B, H, W, step, coef, max_num_iter, _p(cws), _st()),
               "favit_slic_cluster")
    _abi.check(L.favit_slic_connect(_p(lab), _p(ws[0]), _p(ws[1]), _p(out), _p(nreg), B, H, W, min_size, _st()),
               "favit_slic_connect")
    return (out, feat, lab.view(B, H, W), nreg) if stages else out


def slic_stage_times(images, n_segments=16, compactness=0.1, sigma=1.0, max_num_iter=10, min_size_factor=0.5, reps=5):
    """[(stage, ms)] of the three SLIC launches (tools/slic_bench.py)."""
    out = []
    import functools
    B, _, H, W = images.shape
    ys, xs, step = slic_grid(H, W, n_segments)
    dev = images.device
    init = torch.tensor([[y, x] for y in ys for x in xs], dtype=torch.int32, device=dev)
    Kc = init.shape[0]
    coef = int(round((step / float(compactness)) ** 2))
    min_size = int(min_size_factor * (H * W / float(Kc)))
    feat = torch.empty((B, H * W, 4), dtype=torch.int16, device=dev)
    lab = torch.empty((B, H * W), dtype=torch.uint8, device=dev)
    ws = torch.empty((2, B, H * W), dtype=torch.int32, device=dev)
    o = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    nreg = torch.empty(B, dtype=torch.int32, device=dev)
    L = _abi.lib()
    cws = torch.empty(int(L.favit_slic_cluster_workspace(Kc, B)) // 8 + 1, dtype=torch.int64, device=dev)
    fws = torch.empty(int(L.favit_slic_features_workspace(B, H, W)) // 4, dtype=torch.float32, device=dev)
    stages = [
        ("features", lambda: L.favit_slic_features(_p(images), _p(feat), B, H, W, float(sigma), 1, _p(fws), _st())),
        ("cluster", lambda: L.favit_slic_cluster(_p(feat), _p(lab), _p(init), Kc, B, H, W, step, coef, max_num_iter, _p(cws), _st())),
        ("connect", lambda: L.favit_slic_connect(_p(lab), _p(ws[0]), _p(ws[1]), _p(o), _p(nreg), B, H, W, min_size, _st())),
    ]
    for name, fn in stages:
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append((name, e0.elapsed_time(e1) / reps))
    return out


def cross_entropy(logits, labels, grad_scale=None):
    """Returns (loss_rows[B], dlogits or None)."""
    require_gpu(logits, labels)
    if logits.dim() != 2 or logits.dtype != torch.float32 or not logits.is_contiguous():
        raise TypeError("cross_entropy: logits must be a contiguous fp32 [B, C] tensor")
    B, Cn = logits.shape
    if labels.dtype != torch.int64 or tuple(labels.shape) != (B,) or not labels.is_contiguous():
        raise TypeError("cross_entropy: labels must be a contiguous int64 [B] tensor of class indices "
                        "(out-of-range labels, e.g. ignore_index = -100, give a NaN loss row)")
    loss_rows = torch.empty(B, dtype=torch.float32, device=logits.device)
    dlog = torch.empty_like(logits) if grad_scale is not None else None
    _abi.check(_abi.lib().favit_cross_entropy(_p(logits), _p(labels), _p(loss_rows), _p(dlog), B, Cn,
                                              0.0 if grad_scale is None else grad_scale, _st()), "favit_cross_entropy")
    return loss_rows, dlog


def adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, p_lp=None):
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    _abi.check(_abi.lib().favit_adamw(_p(p), _p(g), _p(m), _p(v), _p(p_lp), p.numel(), lr, beta1, beta2, eps, wd, bc1,
                                      bc2, grad_scale, _st()), "favit_adamw")
