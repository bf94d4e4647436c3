"""Host-side mirrors of the reference's L1 model modules (same module names, class names,
constructor/forward signatures, sub-module attribute names and state_dict keys)."""
from . import vit, mhla, vit_mhla, attention, sppp, sppp_mhla, mhla_models  # noqa: F401
