"""Resolves the kernel backend for the model mirrors.

The mirrors can be imported in two ways:
  * as ``<package>.models.vit`` (normal package import), or
  * as top-level ``models.vit`` with the package directory on ``sys.path`` -- the drop-in
    mode in which the reference's ``experiments/*.py`` (``from models.vit import ...``)
    pick these classes up unchanged.
"""
try:
    from .. import functional as F          # noqa: F401
    from .. import kernels as K             # noqa: F401
except ImportError:                          # top-level ``models`` (drop-in mode)
    import importlib
    import os
    import sys

    _pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _parent = os.path.dirname(_pkg_dir)
    if _parent not in sys.path:
        sys.path.insert(0, _parent)
    _pkg = importlib.import_module(os.path.basename(_pkg_dir))
    F = _pkg.functional
    K = _pkg.kernels


def params(module, names):
    """Resolve dotted parameter paths ('attn.qkv.weight', 'mlp.0.bias') on a module."""
    out = []
    for n in names:
        obj = module
        for part in n.split("."):
            obj = getattr(obj, part)
        out.append(obj)
    return out
