"""DeiT with image relative position encoding — host-side mirror of
iRPE/DeiT-with-iRPE/rpe_vision_transformer.py (`RPEAttention` :45-97, `RPEBlock` :100-117,
`VisionTransformer` :120-203) and of the model entry points in rpe_models.py:48-193 /
models.py:152-164, on cream_amd.irpe (HIP `rpe_index` gather / scatter underneath).

Same constructor arguments, parameter names and state-dict keys (`blocks.{i}.attn.rpe_k.
lookup_table_weight`, ...), so the iRPE model-zoo checkpoints load.  timm's `Mlp`,
`PatchEmbed` and `DropPath` (third-party, not vendored in the reference) are restated here
from their published definitions.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import irpe_fused
from .irpe import build_rpe, get_rpe_config


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.drop_prob or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = torch.floor(keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device))
        return x.div(keep) * mask


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


def _matmul_for(x):
    """fp32 device tensors outside autocast (the "within 1e-3 of the reference" configuration) multiply on
    cream_bmm_f32; everything else (host tensors, autocast fallbacks) keeps the framework's matmul."""
    from .autoformer import native_fp32
    return native_fp32.matmul if native_fp32.usable(x) else torch.matmul


def _linear(mod, x):
    """nn.Linear of the layer (qkv / proj, rpe_vision_transformer.py:61-64) — in parity mode on cream_linear_f32_*."""
    from .autoformer import native_fp32
    if native_fp32.usable(x, mod.weight, mod.bias):
        return native_fp32.linear(x, mod.weight, mod.bias, mod.out_features, mod.in_features)
    return mod(x)


class RPEAttention(nn.Module):
    """rpe_vision_transformer.py:45-97.  With s = head_dim^-0.5:
        A = (s q) k^T + rpe_k(s q) + rpe_q(s k)^T ;  P = dropout(softmax(A)) ;  out = P v + rpe_v(P)"""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., rpe_config=None):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.rpe_q, self.rpe_k, self.rpe_v = build_rpe(rpe_config, head_dim=head_dim, num_heads=num_heads)

    def forward(self, x):
        B, N, C = x.shape
        qkv = _linear(self.qkv, x)
        hd = C // self.num_heads
        if irpe_fused.usable(qkv.dtype, qkv.device, hd, N, (self.rpe_q, self.rpe_k, self.rpe_v),
                             dropout_p=self.attn_drop.p if self.training else 0.0):
            # one launch forward, two backward; no (B, H, L, L) tensor exists (csrc/irpe_attn.hip); attn_drop (:86) is
            # applied inside the kernels from a seed
            out = irpe_fused.attention(qkv.view(B, N, 3, self.num_heads, hd), self.scale, self.rpe_q, self.rpe_k,
                                       self.rpe_v, dropout_p=self.attn_drop.p if self.training else 0.0)
            return self.proj_drop(self.proj(out))
        q, k, v = qkv.reshape(B, N, 3, self.num_heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
        q = q * self.scale                                   # (the reference scales q in place, :73)
        mm = _matmul_for(q)                                  # fp32 on the device: the own exact-fp32 kernel, not the library
        attn = mm(q, k.transpose(-2, -1))
        if self.rpe_k is not None:
            attn = attn + self.rpe_k(q)                      # :78-79
        if self.rpe_q is not None:
            attn = attn + self.rpe_q(k * self.scale).transpose(2, 3)     # :82-83
        attn = self.attn_drop(attn.softmax(dim=-1))
        out = mm(attn, v)
        if self.rpe_v is not None:
            out = out + self.rpe_v(attn)                     # :91-92 (post-dropout probabilities)
        return self.proj_drop(_linear(self.proj, out.transpose(1, 2).reshape(B, N, C)))


class RPEBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, rpe_config=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = RPEAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                 attn_drop=attn_drop, proj_drop=drop, rpe_config=rpe_config)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class VisionTransformer(nn.Module):
    """rpe_vision_transformer.py:120-203 (patch input stage only; the hybrid CNN stem of timm is
    outside this path)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=nn.LayerNorm, rpe_config=None):
        super().__init__()
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            RPEBlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                     drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer,
                     rpe_config=rpe_config) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=.02, a=-2., b=2.)
        nn.init.trunc_normal_(self.cls_token, std=.02, a=-2., b=2.)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02, a=-2., b=2.)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def forward_features(self, x):
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1) + self.pos_embed
        x = self.pos_drop(x)
        from . import deit_native
        if deit_native.supported(self.blocks, x):            # bf16 autocast: the whole run of blocks on the own kernels, one node
            x = deit_native.run(self.blocks, x)
        else:
            for blk in self.blocks:
                x = blk(x)
        return self.norm(x)[:, 0]

    def forward(self, x):
        return self.head(self.forward_features(x))


_DEIT = dict(tiny=dict(embed_dim=192, depth=12, num_heads=3), small=dict(embed_dim=384, depth=12, num_heads=6),
             base=dict(embed_dim=768, depth=12, num_heads=12))


def deit_irpe(size='tiny', img_size=224, rpe_on='k', method='product', mode='ctx', ratio=1.9, shared_head=True,
              **kwargs):
    """The registered models of rpe_models.py:48-193 (`deit_{tiny,small,base}_patch16_224_ctx_product_50_
    shared_{k,qk,qkv}`, ...) and `deit_base_patch16_384` with an rpe_config (models.py:152-164): patch 16,
    mlp_ratio 4, qkv_bias, LayerNorm eps 1e-6, one class token (skip=1)."""
    from functools import partial
    cfg = get_rpe_config(ratio=ratio, method=method, mode=mode, shared_head=shared_head, skip=1, rpe_on=rpe_on)
    return VisionTransformer(img_size=img_size, patch_size=16, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), rpe_config=cfg, **{**_DEIT[size], **kwargs})


def deit_tiny_patch16_224_ctx_product_50_shared_k(**kwargs):
    """BASELINE config 1 (rpe_models.py:48-61): 5,755,816 parameters."""
    return deit_irpe('tiny', rpe_on='k', **kwargs)


# The checkpoints the reference's model zoo provides (rpe_models.py:10-19) and the constructors that take them
# (rpe_models.py:48-193): same names, same state-dict keys (tests/golden/irpe_zoo.json, made from the reference's classes).
PROVIDED_CHECKPOINTS = {
    'deit_tiny_patch16_224_ctx_product_50_shared_k': ('tiny', 'k'),
    'deit_small_patch16_224_ctx_product_50_shared_k': ('small', 'k'),
    'deit_small_patch16_224_ctx_product_50_shared_qk': ('small', 'qk'),
    'deit_small_patch16_224_ctx_product_50_shared_qkv': ('small', 'qkv'),
    'deit_base_patch16_224_ctx_product_50_shared_k': ('base', 'k'),
    'deit_base_patch16_224_ctx_product_50_shared_qkv': ('base', 'qkv'),
}


def _load_checkpoint_file(path):
    """DeiT-style files carry an `argparse.Namespace` (`args`) next to `model`: allow that one class through the
    tensors-only unpickler; anything else in the file is refused with a message saying what to do instead."""
    import argparse
    import pickle
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as e:
        raise RuntimeError(f"{path}: holds objects the tensors-only unpickler refuses ({e}); load it yourself "
                           "(torch.load(..., weights_only=False) if you trust the file) and pass the dictionary "
                           "as `checkpoint`") from e


def create_zoo_model(name, checkpoint=None, **kwargs):
    """`register_rpe_model` of rpe_models.py:22-43 without the download: build the named model and, if `checkpoint` (a path
    or the loaded dictionary of a published `<name>.pth`: {'model': state_dict}) is given, load it strictly.  Files are read
    with the tensors-only unpickler."""
    assert name in PROVIDED_CHECKPOINTS, f'Sorry that the checkpoint `{name}` is not provided yet.'       # rpe_models.py:30-31
    size, rpe_on = PROVIDED_CHECKPOINTS[name]
    model = deit_irpe(size, rpe_on=rpe_on, **kwargs)
    if checkpoint is not None:
        ckpt = checkpoint if isinstance(checkpoint, dict) else _load_checkpoint_file(checkpoint)
        model.load_state_dict(ckpt['model'])
    return model


def _zoo_entry(name):
    def build(pretrained=False, checkpoint=None, **kwargs):
        assert not pretrained or checkpoint is not None, "no network here: pass checkpoint=<path or dict> instead of pretrained"
        return create_zoo_model(name, checkpoint, **kwargs)
    build.__name__ = name
    return build


for _n in PROVIDED_CHECKPOINTS:
    if _n not in globals():
        globals()[_n] = _zoo_entry(_n)
