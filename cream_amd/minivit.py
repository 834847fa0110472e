"""Mini-DeiT (weight-shared DeiT with per-repeat iRPE, norms and head transforms) — host-side mirror of
MiniViT/Mini-DeiT/mini_vision_transformer.py (`RepeatedModuleList` :21-33, `MiniAttention` :36-135, `MiniBlock`
:138-165, `RepeatedMiniBlock` :168-189, `VisionTransformer` :192-316) and of the entry points in
mini_deit_models.py:9-59, on cream_amd.irpe / the HIP `rpe_index` operator and the fused iRPE attention.

One `MiniBlock` (qkv, proj, Mlp: the shared weights) is applied `repeated_times` times in a row; what is NOT shared
is selected by the repeat counter: the iRPE tables, the two LayerNorms, the drop-path rate and — with
`use_transform` — the two 1x1 convolutions that mix the heads of the attention map before and after the softmax.
Same constructor arguments, parameter names and state-dict keys as the reference (`blocks.{i}.block.attn.rpe_k.
instances.{r}.lookup_table_weight`, `blocks.{i}.block.norm1.instances.{r}.weight`, `...attn.conv_l.instances.{r}.
weight`), so Mini-DeiT checkpoints load.

Two paths through `MiniAttention.forward`:
  * without head transforms the attention core is exactly `RPEAttention`'s: the fused kernels (csrc/irpe_attn.hip)
    take it under bf16 autocast, with the tables of the current repeat;
  * with head transforms every head's (L, L) map is needed at once (the convolutions contract over heads), so the
    map is formed: q k^T + HIP `rpe_index` gathers, conv_l, softmax, conv_w, P v + bucket sums.

Deviation: with `repeated_times == 1` the reference's `MiniBlock` defines no `norm1` / `norm2` (:146-148) and its
forward raises AttributeError; here that case gets plain LayerNorms (the block is then an `RPEBlock`).
"""
from functools import partial

import torch
import torch.nn as nn

from . import irpe_fused
from .irpe import build_rpe, get_rpe_config
from .rpe_attention import DropPath, Mlp, PatchEmbed


class RepeatedModuleList(nn.Module):
    """mini_vision_transformer.py:21-33: `repeated_times` instances, the one of the current repeat is called."""

    def __init__(self, instances, repeated_times):
        super().__init__()
        assert len(instances) == repeated_times
        self.instances = nn.ModuleList(instances)
        self.repeated_times = repeated_times
        self._repeated_id = 0

    def current(self):
        return self.instances[self._repeated_id]

    def forward(self, *args, **kwargs):
        return self.instances[self._repeated_id](*args, **kwargs)

    def extra_repr(self):
        return f'repeated_times={self.repeated_times}'


def _current(m):
    return m.current() if isinstance(m, RepeatedModuleList) else m


class MiniAttention(nn.Module):
    """mini_vision_transformer.py:36-135."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., rpe_config=None,
                 repeated_times=1, use_transform=False):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        per_repeat = [build_rpe(rpe_config, head_dim=head_dim, num_heads=num_heads) for _ in range(repeated_times)]
        rpe_q, rpe_k, rpe_v = zip(*per_repeat)
        self.rpe_q = RepeatedModuleList(rpe_q, repeated_times) if rpe_q[0] is not None else None
        self.rpe_k = RepeatedModuleList(rpe_k, repeated_times) if rpe_k[0] is not None else None
        self.rpe_v = RepeatedModuleList(rpe_v, repeated_times) if rpe_v[0] is not None else None
        if use_transform:
            conv = lambda: nn.Conv2d(num_heads, num_heads, kernel_size=1, bias=False)          # noqa: E731
            self.conv_l = RepeatedModuleList([conv() for _ in range(repeated_times)], repeated_times)
            self.conv_w = RepeatedModuleList([conv() for _ in range(repeated_times)], repeated_times)
        else:
            self.conv_l = self.conv_w = None

    def init_weights(self):
        for m in (self.conv_l, self.conv_w):
            if m is not None:
                for c in m.instances:
                    nn.init.trunc_normal_(c.weight, std=.02, a=-2., b=2.)

    @staticmethod
    def _mix_heads(conv, attn):
        """The 1x1 convolution over the head axis of (B, H, L, L) as one contraction (no im2col, no bias: :76)."""
        return torch.einsum('oh,bhij->boij', conv.weight[:, :, 0, 0].to(attn.dtype), attn)

    def forward(self, x):
        B, N, C = x.shape
        hd = C // self.num_heads
        qkv = self.qkv(x)
        rq, rk, rv = (_current(m) if m is not None else None for m in (self.rpe_q, self.rpe_k, self.rpe_v))
        if self.conv_l is None and irpe_fused.usable(qkv.dtype, qkv.device, hd, N, (rq, rk, rv),
                                                     dropout_p=self.attn_drop.p if self.training else 0.0):
            out = irpe_fused.attention(qkv.view(B, N, 3, self.num_heads, hd), self.scale, rq, rk, rv,
                                       dropout_p=self.attn_drop.p if self.training else 0.0)
            return self.proj_drop(self.proj(out))
        q, k, v = qkv.reshape(B, N, 3, self.num_heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
        q = q * self.scale                                   # :88
        attn = q @ k.transpose(-2, -1)
        if rk is not None:
            attn = attn + rk(q)                              # :93-94
        if rq is not None:
            attn = attn + rq(k * self.scale).transpose(2, 3)             # :97-98
        if self.conv_l is not None:
            attn = self._mix_heads(self.conv_l.current(), attn)          # :100-101
        attn = attn.softmax(dim=-1)
        if self.conv_w is not None:
            attn = self._mix_heads(self.conv_w.current(), attn)          # :105-106
        attn = self.attn_drop(attn)
        out = attn @ v
        if rv is not None:
            out = out + rv(attn)                             # :113-114
        return self.proj_drop(self.proj(out.transpose(1, 2).reshape(B, N, C)))


class MiniBlock(nn.Module):
    """mini_vision_transformer.py:138-165."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_paths=(0.,), act_layer=nn.GELU, norm_layer=nn.LayerNorm, rpe_config=None, repeated_times=1,
                 use_transform=False):
        super().__init__()
        assert len(drop_paths) == repeated_times
        if repeated_times > 1:
            self.norm1 = RepeatedModuleList([norm_layer(dim) for _ in range(repeated_times)], repeated_times)
            self.norm2 = RepeatedModuleList([norm_layer(dim) for _ in range(repeated_times)], repeated_times)
        else:
            self.norm1, self.norm2 = norm_layer(dim), norm_layer(dim)
        self.attn = MiniAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                                  proj_drop=drop, rpe_config=rpe_config, repeated_times=repeated_times,
                                  use_transform=use_transform)
        self.drop_paths = nn.ModuleList([DropPath(p) if p > 0. else nn.Identity() for p in drop_paths])
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self._repeated_id = 0

    def forward(self, x):
        drop_path = self.drop_paths[self._repeated_id]
        x = x + drop_path(self.attn(self.norm1(x)))
        return x + drop_path(self.mlp(self.norm2(x)))


class RepeatedMiniBlock(nn.Module):
    """mini_vision_transformer.py:168-189.  The reference re-applies a closure over every submodule for every repeat;
    here the modules that read the counter are collected once."""

    def __init__(self, repeated_times, **kwargs):
        super().__init__()
        self.repeated_times = repeated_times
        self.block = MiniBlock(repeated_times=repeated_times, **kwargs)
        self._counted = None

    def _set_repeat(self, r):
        if self._counted is None:
            self._counted = [m for m in self.block.modules() if isinstance(m, (RepeatedModuleList, MiniBlock))]
        for m in self._counted:
            m._repeated_id = r

    def forward(self, x):
        for r in range(self.repeated_times):
            self._set_repeat(r)
            x = self.block(x)
        return x

    def extra_repr(self):
        return f'repeated_times={self.repeated_times}'


class MiniVisionTransformer(nn.Module):
    """mini_vision_transformer.py:192-316 (patch input stage; `use_cls_token=False` pools the tokens instead)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 norm_layer=nn.LayerNorm, rpe_config=None, use_cls_token=True, repeated_times=1, use_transform=False):
        super().__init__()
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if use_cls_token else None
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1 if use_cls_token else n, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]
        assert depth % repeated_times == 0
        kw = dict(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, norm_layer=norm_layer, rpe_config=rpe_config,
                  use_transform=use_transform)
        blocks = []
        for i in range(depth // repeated_times):
            if repeated_times > 1:
                blocks.append(RepeatedMiniBlock(repeated_times=repeated_times,
                                                drop_paths=dpr[i * repeated_times:(i + 1) * repeated_times], **kw))
            else:
                blocks.append(MiniBlock(drop_paths=[dpr[i]], **kw))
        self.blocks = nn.ModuleList(blocks)
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=.02, a=-2., b=2.)
        if self.cls_token is not None:
            nn.init.trunc_normal_(self.cls_token, std=.02, a=-2., b=2.)
        self.apply(self._init_weights)
        for m in self.modules():
            if isinstance(m, MiniAttention):
                m.init_weights()

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
        if self.cls_token is not None:
            x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        x = self.pos_drop(x + self.pos_embed)
        for blk in self.blocks:
            x = blk(x)
        x = self.norm(x)
        return x[:, 0] if self.cls_token is not None else x

    def forward(self, x):
        x = self.forward_features(x)
        if self.cls_token is None:
            x = x.mean(1)                                    # AdaptiveAvgPool1d(1) over the tokens (:311-313)
        return self.head(x)


_DEIT = dict(tiny=dict(embed_dim=192, depth=12, num_heads=3), small=dict(embed_dim=384, depth=12, num_heads=6),
             base=dict(embed_dim=768, depth=12, num_heads=12))


def mini_deit(size='tiny', img_size=224, **kwargs):
    """`mini_deit_{tiny,small,base}_patch16_224`, `mini_deit_base_patch16_384` (mini_deit_models.py:9-59): rpe on k,
    product method, contextual, shared heads, skip = 0 (no class token), two repeats, head transforms on."""
    cfg = get_rpe_config(ratio=1.9, method='product', mode='ctx', shared_head=True, skip=0, rpe_on='k')
    kw = dict(img_size=img_size, patch_size=16, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
              rpe_config=cfg, use_cls_token=False, repeated_times=2, use_transform=True)
    kw.update(_DEIT[size])
    kw.update(kwargs)
    return MiniVisionTransformer(**kw)
