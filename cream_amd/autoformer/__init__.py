"""AutoFormer weight-entangled supernet (reference: AutoFormer/model, AutoFormer/supernet_engine.py)."""
from .modules import (AttentionSuper, LayerNormSuper, LinearSuper, PatchembedSuper,  # noqa: F401
                      RelativePosition2D_super, qkv_super)
from .supernet import TransformerEncoderLayer, Vision_TransformerSuper, gelu  # noqa: F401
