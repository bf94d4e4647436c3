#!/usr/bin/env python3
"""Deterministic hunt for reads of memory nobody wrote.  `install()` closes, for one process, the classes of stale
memory a kernel can consume without any allocator noticing:

* **LDS**: in front of EVERY libfavit call a kernel fills all 160 KiB of LDS of every CU with a NaN pattern
  (tools/poison_lds.hip -> tools/libpoison.so), in stream order on the calling thread's current stream — also inside
  graph captures, so replays are poisoned too.  A kernel that feeds an LDS row it never staged into an MFMA or a
  reduction (ragged tiles, halo rows, dump slots, rows multiplied by a zero weight: NaN * 0 = NaN) fails every time.
* **the cached split-K slab workspace** (`kernels._GROUPED_WS`): NaN-filled in front of every grouped weight-gradient
  launch, so a slab element no split writes reaches the reduction as NaN instead of as the previous launch's value.
* **fresh allocations**: `torch.empty`, `empty_like`, `empty_strided`, `Tensor.new_empty` fill their result with NaN
  (floating types) or 0xFF bytes (integers; as fp32 / bf16 that is a NaN as well).

What it cannot see: out-of-bounds reads into live neighbouring tensors, registers, and ATen-internal allocations.

Use: `FAVIT_POISON=1 python -m pytest tests -m gpu -x -q` (tests/conftest.py calls install()), or
`python tools/poison.py cfg1 cfg3 cfg5` (eager and graph-replayed training steps of bench configurations; exits
non-zero and names the first non-finite tensor).  `python tools/poison.py --census` prints how many distinct CUs one
poison launch covered.
"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

PATTERN = 0x7FC07FC0
_NO_LAUNCH = {"favit_abi_version", "favit_strerror", "favit_set_dropout_epoch", "favit_gemm_last_kernel",
              "favit_gemm_grouped_tn_workspace", "favit_mhla_attn_lse_supported", "favit_slic_features_workspace",
              "favit_slic_cluster_workspace"}
_state = {"installed": False, "calls": 0}


def _poison_lib():
    path = os.path.join(ROOT, "tools", "libpoison.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} missing: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/poison_lds.hip -o tools/libpoison.so")
    lib = C.CDLL(path)
    lib.favit_tool_poison_lds.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
    lib.favit_tool_poison_lds.restype = C.c_int
    return lib


def _fill(t):
    if t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64) and t.is_contiguous():
            t.view(torch.uint8).fill_(0xFF)
    return t


class _PoisonedLib:
    """Stands in for the ctypes library object of _abi.lib(): same attributes, every launching entry point preceded by
    the LDS poison kernel (and the slab workspace fill) on the current stream."""

    def __init__(self, real, plib, kernels):
        self._real, self._plib, self._K = real, plib, kernels
        self._cache = {}

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name in _NO_LAUNCH or not name.startswith("favit_"):
            return fn
        w = self._cache.get(name)
        if w is None:
            def w(*a, _fn=fn, _name=name):
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                if _name == "favit_gemm_grouped_tn_ws":
                    ws = self._K._GROUPED_WS.get(torch.cuda.current_device())
                    if ws is not None:
                        ws[: ws.numel() // 4 * 4].view(torch.float32).fill_(float("nan"))
                rc = self._plib.favit_tool_poison_lds(st, PATTERN, None)
                if rc != 0:
                    raise RuntimeError(f"poison launch failed ({rc})")
                _state["calls"] += 1
                return _fn(*a)
            self._cache[name] = w
        return w


def install(lds=True, empty=True):
    """Idempotent.  Returns the package."""
    pkg = importlib.import_module("focused-attention-vit_amd")
    if _state["installed"]:
        return pkg
    if empty:
        _e, _el, _es, _ne = torch.empty, torch.empty_like, torch.empty_strided, torch.Tensor.new_empty
        torch.empty = lambda *a, **k: _fill(_e(*a, **k))
        torch.empty_like = lambda *a, **k: _fill(_el(*a, **k))
        torch.empty_strided = lambda *a, **k: _fill(_es(*a, **k))
        torch.Tensor.new_empty = lambda self, *a, **k: _fill(_ne(self, *a, **k))
    if lds:
        real = pkg._abi.lib()
        pkg._abi._lib = _PoisonedLib(real, _poison_lib(), pkg.kernels)
    _state["installed"] = True
    return pkg


def first_nonfinite(named):
    """Name and count of the first tensor of `named` ((name, tensor) pairs) that holds a non-finite value, else None."""
    for n, t in named:
        if t is None or not torch.is_tensor(t) or not t.is_floating_point():
            continue
        bad = ~torch.isfinite(t)
        if bool(bad.any()):
            return n, int(bad.sum()), tuple(t.shape)
    return None


def _run_config(pkg, cfg, steps):
    import numpy as np
    import bench
    c = bench.CONFIGS[cfg]
    dev = torch.device("cuda", 0)
    B = int(os.environ.get("BATCH", c["batch"]))
    g = torch.Generator(device=dev).manual_seed(1234)
    images = torch.randn(B, 3, c["img"], c["img"], device=dev, generator=g)
    labels = torch.randint(0, c["classes"], (B,), device=dev, generator=g)

    def fresh():
        torch.manual_seed(1234)
        m = bench.build_model(pkg, cfg, dev, float(os.environ.get("DROPOUT", "0"))).train()
        if cfg in ("cfg3", "cfg5"):
            segs_np = bench.synthetic_label_maps(8, 224, 16, seed=100)
            m.segmentation.set_label_maps(torch.from_numpy(np.stack([segs_np[i % 8] for i in range(B)])).to(dev))
            m.assume_num_tokens = 16
        o = pkg.train.FusedAdamW(pkg.train.param_groups(m, lr=1e-4), lr=1e-4, weight_decay=0.05)
        return m, o

    def check(tag, s, m, o, loss):
        named = [("loss", loss)]
        for gi, g_ in enumerate(o.groups):
            named += [(f"group{gi}.flat_g", g_["flat"].flat_g), (f"group{gi}.flat_p", g_["flat"].flat_p),
                      (f"group{gi}.m", g_["m"]), (f"group{gi}.v", g_["v"])]
        named += [(n, p.grad) for n, p in m.named_parameters() if p.grad is not None]
        named += list(m.named_parameters())
        torch.cuda.synchronize()
        bad = first_nonfinite(named)
        print(f"{cfg} {tag} step {s}: loss {float(loss):.5f}  first non-finite: {bad}", flush=True)
        if bad is not None:
            t = dict(named)[bad[0]].detach().flatten()
            idx = (~torch.isfinite(t)).nonzero().flatten()
            runs = (idx[1:] != idx[:-1] + 1).nonzero().flatten() + 1
            starts = [int(idx[0])] + [int(idx[r]) for r in runs[:12]]
            print(f"    indices {int(idx[0])}..{int(idx[-1])}, {len(runs) + 1} contiguous run(s), run starts {starts}, "
                  f"values {t[idx[:4]].tolist()}", flush=True)
            for gi, g_ in enumerate(o.groups):
                if bad[0].startswith(f"group{gi}."):
                    names = {id(p_): n for n, p_ in m.named_parameters()}
                    flat = g_["flat"]
                    hit = {}
                    for p_, off in zip(flat.params, flat.offsets):
                        c = int(((idx >= off) & (idx < off + p_.numel())).sum())
                        if c:
                            hit[names[id(p_)]] = (c, p_.numel(), off)
                    print(f"    by parameter (count, numel, flat offset): {hit}", flush=True)
        return bad is None

    def grads(tag, s, m):
        """Between backward and AdamW: parameters whose gradient is non-finite or absurdly large (a huge finite gradient
        overflows AdamW's second moment and silently freezes the parameter)."""
        torch.cuda.synchronize()
        bad = []
        for n, p_ in m.named_parameters():
            if p_.grad is None:
                continue
            g_ = p_.grad
            nf = int((~torch.isfinite(g_)).sum())
            big = int((g_.abs() > 1e6).sum())
            if nf or big:
                idx = ((~torch.isfinite(g_)) | (g_.abs() > 1e6)).flatten().nonzero().flatten()
                bad.append(f"{n}{tuple(g_.shape)}: {nf} non-finite, {big} |g|>1e6, first flat idx {idx[:6].tolist()} last {int(idx[-1])} "
                           f"sample {g_.flatten()[idx[:3]].tolist()}")
        for b in bad:
            print(f"{cfg} {tag} step {s} GRADIENT {b}", flush=True)
        return not bad

    ok = True
    m, o = fresh()
    for s in range(steps):
        o.zero_grad()
        loss = pkg.train.cross_entropy(m(images), labels)
        loss.backward()
        ok &= grads("eager", s, m)
        o.step()
        ok &= check("eager", s, m, o, loss)
    pkg.functional.clear_lp_mirrors()
    m, o = fresh()
    gs = pkg.train.GraphedStep(m, o, images, labels)
    for s in range(steps):
        for g_ in gs.graphs:
            g_.replay()
        ok &= grads("graph", s, m)
        o.step()
        ok &= check("graph", s, m, o, gs.loss)
    del gs
    pkg.functional.clear_lp_mirrors()
    return ok


def main():
    if "--census" in sys.argv:
        plib = _poison_lib()
        cen = torch.zeros(512, dtype=torch.int32, device="cuda")
        plib.favit_tool_poison_lds(C.c_void_p(torch.cuda.current_stream().cuda_stream), PATTERN, C.c_void_p(cen.data_ptr()))
        torch.cuda.synchronize()
        print(f"distinct (xcc, cu) ids over one poison launch of 512 workgroups: {len(set(cen.tolist()))}")
        return 0
    pkg = install(lds=os.environ.get("POISON_LDS", "1") != "0", empty=os.environ.get("POISON_EMPTY", "1") != "0")
    pkg.set_compute_dtype(os.environ.get("DTYPE", "bf16"))
    cfgs = [a for a in sys.argv[1:] if a.startswith("cfg")] or ["cfg1", "cfg3"]
    steps = int(os.environ.get("STEPS", "4"))
    ok = True
    for cfg in cfgs:
        ok &= _run_config(pkg, cfg, steps)
    print(f"poison launches: {_state['calls']}; {'CLEAN' if ok else 'NON-FINITE VALUES FOUND'}", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
