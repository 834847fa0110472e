"""Fused attention + 2-D relative position bias (HIP, cream_attn_rpe2d_fwd/bwd)."""


def available():
    return False


def supported(qkv, dropout_p):
    return False


def attention_rpe2d_fused(qkv, tkv, tkh, tvv, tvh, iv, ih, scale):
    raise NotImplementedError("fused attention kernels are not built yet")
