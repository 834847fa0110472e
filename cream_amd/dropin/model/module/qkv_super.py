"""Drop-in for AutoFormer/model/module/qkv_super.py (re-export; implementation in cream_amd.autoformer.modules)."""
from cream_amd.autoformer.modules import qkv_super  # noqa: F401
