"""Drop-in for AutoFormer/model/module/multihead_super.py (re-export)."""
from cream_amd.autoformer.modules import AttentionSuper, RelativePosition2D_super  # noqa: F401
