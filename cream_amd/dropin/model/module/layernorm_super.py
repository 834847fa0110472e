"""Drop-in for AutoFormer/model/module/layernorm_super.py (re-export; implementation in cream_amd.autoformer.modules)."""
from cream_amd.autoformer.modules import LayerNormSuper  # noqa: F401
