"""focused-attention-vit_amd: MI355X-native (gfx950) implementation of the focused-attention
ViT encoder forward/backward hot path of zser092/Focused-Attention-ViT.

The directory name carries a hyphen (it is the project's package name), so import it with
``importlib.import_module("focused-attention-vit_amd")`` -- or put this directory on
``sys.path`` to get drop-in ``models.vit`` / ``models.mhla`` / ``models.vit_mhla`` /
``models.sppp`` / ``models.sppp_mhla`` / ``models.attention`` modules with the reference's
class names, constructor/forward signatures and state_dict keys.

The math runs in hand-written HIP kernels (``csrc/``) reached through the C ABI of
``include/favit.h``; there is no CPU or PyTorch-op fallback.
"""
from . import _abi
from .functional import (get_compute_dtype, get_compute_mode, set_compute_dtype, set_side_stream, set_direct_grads,
                         invalidate_weight_cache)
from . import kernels, functional, models
from . import dp, train
from . import data, harness, streams

__all__ = ["set_compute_dtype", "get_compute_dtype", "get_compute_mode", "invalidate_weight_cache", "kernels", "functional", "models", "dp", "train", "data", "harness", "_abi"]
