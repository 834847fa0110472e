"""TinyCLIP's two towers — host-side mirror of TinyCLIP/src/open_clip/model.py for the dense (mask-free) models of the
manual weight-inheritance recipe: `ResidualAttentionBlock` :208-315, `Transformer` :342-427, `VisualTransformer`
:442-543, `ImageEncoder` :597-675, `TextEncoder` :682-818, `LogitScale` :847-853, `CLIPBase` / `CLIP` :874-1112.

Same constructor configuration (the JSON files of open_clip/model_configs), parameter names and state-dict keys
(`_image_encoder.visual.transformer.resblocks.{i}.attn.in_proj_weight`, `_text_encoder.token_embedding.weight`,
`_logit_scale.logit_scale`, ...), so TinyCLIP / OpenCLIP checkpoints in the reference's "new" layout load.

What is different is the execution: the towers run batch-first (the reference permutes to sequence-first for
`nn.MultiheadAttention` and back), and the attention core of a block whose heads are 64 wide runs — under bf16 autocast
on the device — on the fused flash-style kernels of csrc/irpe_attn.hip (`irpe_fused.attention` without any relative
position term: nothing of size L^2 reaches HBM, forward one launch, backward two).  Under bf16 autocast on the device the
whole run of blocks of a tower — image AND text (causal mask inside the attention kernels) — is one autograd node on the
own GEMM / LayerNorm / attention kernels (cream_amd.tinyclip.native); every fp32 / CPU call uses the composed form.

Not mirrored (outside the affinity-mimicking step of BASELINE config 5): the learnable pruning masks (`l0module.py`,
`hidden_z` / `heads_z` / ... arguments and `prune()`), the ResNet / timm image towers, gradient checkpointing.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from .. import irpe_fused

NATIVE_TOWERS = os.environ.get('CREAM_TINYCLIP_NATIVE', '1') != '0'     # image towers on cream_amd.tinyclip.native


class LayerNorm(nn.LayerNorm):
    """model.py:40-68 without the mask branch (torch's layer_norm handles low-precision inputs under autocast)."""

    def forward(self, x):
        return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps).to(x.dtype)


class QuickGELU(nn.Module):
    def forward(self, x):                                       # model.py:102-105
        return x * torch.sigmoid(1.702 * x)


class Mlp(nn.Module):
    def __init__(self, d_model, mlp_width, act_layer=nn.GELU):   # model.py:108-137
        super().__init__()
        self.c_fc = nn.Linear(d_model, mlp_width)
        self.gelu = act_layer()
        self.c_proj = nn.Linear(mlp_width, d_model)

    def forward(self, x):
        return self.c_proj(self.gelu(self.c_fc(x)))


class MultiheadAttention(nn.Module):
    """The parameters of `nn.MultiheadAttention(d_model, n_head)` (packed `in_proj_weight` / `in_proj_bias`, `out_proj`)
    with a batch-first self-attention forward."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, x, attn_mask=None):
        """x (N, L, D); attn_mask additive (L, L) or None."""
        N, L, D = x.shape
        H, hd = self.num_heads, self.head_dim
        qkv = F.linear(x, self.in_proj_weight, self.in_proj_bias)
        if attn_mask is None and irpe_fused.usable(qkv.dtype, qkv.device, hd, L, (None, None, None), False):
            out = irpe_fused.attention(qkv.view(N, L, 3, H, hd), hd ** -0.5, None, None, None)       # (N, L, D)
            return self.out_proj(out)
        q, k, v = qkv.view(N, L, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)                            # (N, H, L, hd)
        attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
        if attn_mask is not None:
            attn = attn + attn_mask.to(attn.dtype)
        out = attn.softmax(dim=-1) @ v
        return self.out_proj(out.transpose(1, 2).reshape(N, L, D))


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, mlp_ratio=4.0, act_layer=nn.GELU):      # model.py:208-236
        super().__init__()
        self.ln_1 = LayerNorm(d_model)
        self.attn = MultiheadAttention(d_model, n_head)
        self.ln_attn = nn.Identity()
        self.ln_2 = LayerNorm(d_model)
        self.mlp = Mlp(d_model, int(d_model * mlp_ratio), act_layer)

    def forward(self, x, attn_mask=None):                                       # model.py:285-315
        x = x + self.attn(self.ln_1(x), attn_mask)
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, mlp_ratio=4.0, act_layer=nn.GELU):  # model.py:342-359
        super().__init__()
        self.width, self.layers, self.num_heads, self.head_dim, self.mlp_ratio = width, layers, heads, width // heads, mlp_ratio
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads, mlp_ratio, act_layer) for _ in range(layers)])

    def forward(self, x, attn_mask=None, causal=False):
        """`causal=True`: the caller states that attn_mask is the upper-triangular -inf mask (the text tower's buffer)."""
        if x.is_cuda and NATIVE_TOWERS:
            from . import native
            if native.supported(self, x, attn_mask, causal):   # bf16 autocast, heads of 64: the own kernels end to end
                return native.tower(self, x, causal and attn_mask is not None)
        for blk in self.resblocks:
            x = blk(x, attn_mask)
        return x


class VisualTransformer(nn.Module):
    def __init__(self, image_size, patch_size, width, layers, heads, mlp_ratio, output_dim, act_layer=nn.GELU):
        super().__init__()                                                      # model.py:442-482
        self.image_size, self.patch_size = (image_size, image_size), (patch_size, patch_size)
        self.grid_size = (image_size // patch_size, image_size // patch_size)
        self.output_dim, self.embed_dim, self.layers, self.head_dim = output_dim, width, layers, width // heads
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid_size[0] * self.grid_size[1] + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, mlp_ratio, act_layer=act_layer)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def patches(self, x):
        """`conv1` (kernel = stride = patch, no bias, model.py:462-463,503-507) as ONE GEMM over the unfolded patches:
        (N * grid^2, 3 * p * p) x (3 * p * p, width).  Against the framework's convolution route (which first spends its
        warm-up steps searching, naive kernels included) the steady-state step is 1 % faster (same-box A/B, 81.0 -> 80.3 ms)."""
        N, C, Hh, Ww = x.shape
        p = self.patch_size[0]
        gh, gw = Hh // p, Ww // p
        x = x.reshape(N, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(N, gh * gw, C * p * p)
        return F.linear(x, self.conv1.weight.reshape(self.conv1.out_channels, -1))

    def forward(self, x):                                                       # model.py:493-535
        x = self.patches(x)                                                     # (N, grid^2, width)
        cls = self.class_embedding.to(x.dtype).expand(x.shape[0], 1, -1)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype)
        x = self.transformer(self.ln_pre(x))
        return self.ln_post(x[:, 0, :]) @ self.proj


class ImageEncoder(nn.Module):
    def __init__(self, embed_dim, vision_cfg, quick_gelu):                      # model.py:597-631 (ViT branch)
        super().__init__()
        c = dict(layers=12, width=768, head_width=64, mlp_ratio=4.0, patch_size=16, image_size=224)
        c.update(vision_cfg)
        if c.get("timm_model_name") or isinstance(c["layers"], (tuple, list)):
            raise NotImplementedError("ImageEncoder: only the ViT tower is on this path (model.py:603-622 are the timm / ResNet towers)")
        self.visual = VisualTransformer(c["image_size"], c["patch_size"], c["width"], c["layers"], c["width"] // c["head_width"],
                                        c["mlp_ratio"], embed_dim, act_layer=QuickGELU if quick_gelu else nn.GELU)
        self.l0_module = None

    def forward(self, image, normalized=False):
        f = self.visual(image)
        return F.normalize(f, dim=-1) if normalized else f


class TextEncoder(nn.Module):
    def __init__(self, embed_dim, text_cfg, quick_gelu):                        # model.py:682-718
        super().__init__()
        c = dict(context_length=77, vocab_size=49408, width=512, heads=8, layers=12)
        c.update(text_cfg)
        self.context_length, self.vocab_size = c["context_length"], c["vocab_size"]
        self.transformer = Transformer(c["width"], c["layers"], c["heads"], act_layer=QuickGELU if quick_gelu else nn.GELU)
        self.token_embedding = nn.Embedding(c["vocab_size"], c["width"])
        self.positional_embedding = nn.Parameter(torch.empty(self.context_length, c["width"]))
        self.ln_final = LayerNorm(c["width"])
        self.text_projection = nn.Parameter(torch.empty(c["width"], embed_dim))
        self.register_buffer("attn_mask", self.build_attention_mask(), persistent=False)
        self.l0_module = None
        self.init_parameters()

    def init_parameters(self):                                                  # model.py:737-754
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        t = self.transformer
        proj_std, attn_std, fc_std = (t.width ** -0.5) * ((2 * t.layers) ** -0.5), t.width ** -0.5, (2 * t.width) ** -0.5
        for blk in t.resblocks:
            nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(blk.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=t.width ** -0.5)

    def build_attention_mask(self):                                             # model.py:756-762 (causal, additive)
        return torch.full((self.context_length, self.context_length), float("-inf")).triu_(1)

    def forward(self, text, normalized=False):                                  # model.py:764-805
        x = self.token_embedding(text) + self.positional_embedding
        x = self.ln_final(self.transformer(x, attn_mask=self.attn_mask[:x.shape[1], :x.shape[1]], causal=True))
        x = x[torch.arange(x.shape[0], device=x.device), text.argmax(dim=-1)] @ self.text_projection   # the eot token
        return F.normalize(x, dim=-1) if normalized else x


class LogitScale(nn.Module):
    def __init__(self):                                                         # model.py:847-853
        super().__init__()
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    def forward(self, dummy=None):
        return self.logit_scale


class CLIP(nn.Module):
    """model.py:874-1112: `forward(image, text, normalized=True)` -> (image features, text features, exp(logit scale));
    a tower whose input is None is skipped (:990-1001)."""

    def __init__(self, embed_dim, vision_cfg, text_cfg, quick_gelu=False):
        super().__init__()
        self._image_encoder = ImageEncoder(embed_dim, vision_cfg, quick_gelu)
        self._text_encoder = TextEncoder(embed_dim, text_cfg, quick_gelu)
        self._logit_scale = LogitScale()
        # per-tower precision (model.py:883-896; train.py:91-99 passes get_autocast(args.image_precision / text_precision /
        # logit_precision)): context-manager FACTORIES, entered around each tower
        from contextlib import nullcontext
        self.image_autocast = self.text_autocast = self.logit_autocast = nullcontext

    def set_autocast(self, image_autocast, text_autocast, logit_autocast):      # model.py:893-896
        self.image_autocast, self.text_autocast, self.logit_autocast = image_autocast, text_autocast, logit_autocast

    visual = property(lambda self: self._image_encoder.visual)
    transformer = property(lambda self: self._text_encoder.transformer)
    logit_scale = property(lambda self: self._logit_scale.logit_scale)

    def encode_image(self, image, normalized=False):                            # model.py:1003-1005
        with self.image_autocast():
            return self._image_encoder(image, normalized=normalized)

    def encode_text(self, text, normalized=False):                              # model.py:1007-1009
        with self.text_autocast():
            return self._text_encoder(text, normalized=normalized)

    def forward(self, image, text, normalized=True):                            # model.py:990-1001
        fi = ft = None
        if image is not None:
            with self.image_autocast():
                fi = self._image_encoder(image, normalized=normalized)
        if text is not None:
            with self.text_autocast():
                ft = self._text_encoder(text, normalized=normalized)
        with self.logit_autocast():
            scale = self._logit_scale()
        return fi, ft, scale.exp()

    def load_state_dict(self, state_dict, strict=True):                         # model.py:1049-1070
        return super().load_state_dict(convert_to_new_checkpoint(state_dict), strict=strict)

    def lock_image_tower(self):                                                 # model.py:1015-1021
        for p in self._image_encoder.parameters():
            p.requires_grad = False

    def lock_text_tower(self):
        for p in self._text_encoder.parameters():
            p.requires_grad = False


def convert_to_new_checkpoint(state_dict):
    """Any of the layouts TinyCLIP / OpenCLIP checkpoints come in -> the `_image_encoder. / _text_encoder. /
    _logit_scale.` layout of this class (model.py:1115-1157 with `used_ddp=False`, plus the `.module` strip of
    `CLIPBase.load_state_dict`, :1066-1070):
      * the new layout saved from DDP-wrapped towers (`_image_encoder.module.visual...`): the `module` level is removed;
      * the old single-module layout (`visual.*`, `logit_scale`, text keys at the root), optionally under `module.`."""
    if '_logit_scale.module.logit_scale' in state_dict:
        out = {}
        for k, v in state_dict.items():
            sp = k.split('.')
            assert sp[1] == 'module', k
            out['.'.join(sp[:1] + sp[2:])] = v
        return out
    if '_logit_scale.logit_scale' in state_dict:
        return dict(state_dict)
    if 'module.logit_scale' in state_dict:
        state_dict = {k[len('module.'):]: v for k, v in state_dict.items()}
    if 'logit_scale' not in state_dict:
        return dict(state_dict)
    out = {}
    for k, v in state_dict.items():
        if k.startswith('visual.'):
            out['_image_encoder.' + k] = v
        elif k == 'logit_scale':
            out['_logit_scale.logit_scale'] = v
        else:
            out['_text_encoder.' + k] = v
    return out


# open_clip/model_configs/*.json of the configurations BASELINE config 5 names
MODEL_CONFIGS = OrderedDict([
    ("TinyCLIP-ViT-39M-16-Text-19M", dict(embed_dim=512, vision_cfg=dict(image_size=224, layers=12, width=512, patch_size=16),
                                          text_cfg=dict(context_length=77, vocab_size=49408, width=512, heads=8, layers=6))),
    ("ViT-B-16", dict(embed_dim=512, vision_cfg=dict(image_size=224, layers=12, width=768, patch_size=16),
                      text_cfg=dict(context_length=77, vocab_size=49408, width=512, heads=8, layers=12))),
])


def create_model(name, **overrides):
    """factory.py `create_model` for the two configurations above (no pretrained download: there is no network)."""
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in MODEL_CONFIGS[name].items()}
    for k, v in overrides.items():
        if isinstance(v, dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    return CLIP(**cfg)


def n_params(m):
    return sum(p.numel() for p in m.parameters())


__all__ = ["CLIP", "ImageEncoder", "TextEncoder", "VisualTransformer", "Transformer", "ResidualAttentionBlock", "LayerNorm",
           "QuickGELU", "MODEL_CONFIGS", "create_model", "n_params", "convert_to_new_checkpoint"]
