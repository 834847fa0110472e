"""Drop-in for AutoFormer/model/module/embedding_super.py (re-export; implementation in cream_amd.autoformer.modules)."""
from cream_amd.autoformer.modules import PatchembedSuper  # noqa: F401
