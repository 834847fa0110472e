"""Drop-in for AutoFormer/model/module/Linear_super.py (re-export; implementation in cream_amd.autoformer.modules)."""
from cream_amd.autoformer.modules import LinearSuper  # noqa: F401
