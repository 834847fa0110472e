"""Vision_TransformerSuper / TransformerEncoderLayer — host-side mirror of
AutoFormer/model/supernet_transformer.py (same constructor arguments, parameter names,
`set_sample_config` protocol and counters), built on cream_amd.autoformer.modules.

Semantics kept from the reference:
  * pre-norm blocks; layers >= layer_num are identity (supernet_transformer.py:259-260);
  * head dim is 64 under change_qkv: sample_q_embed_dim = heads * 64 (:243);
  * gelu runs in fp32 and casts back (:14-16);
  * `F.dropout` after attention uses sample_attn_dropout, the Mlp uses sample_dropout (:268,278-280);
  * drop-path probabilities are a linspace over the SUPER depth (:51-56);
  * gp=True: mean over tokens 1.. AFTER the final LayerNorm (:161-165).
"""
import numpy as np
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import block as _block
from .modules import AttentionSuper, LayerNormSuper, LinearSuper, PatchembedSuper, _trunc_normal_


def _bf16_autocast(x):
    return (x.is_cuda and torch.is_autocast_enabled('cuda')
            and torch.get_autocast_dtype('cuda') == torch.bfloat16)


def gelu(x):
    return F.gelu(x.float()).type_as(x)


def calc_dropout(dropout, sample_embed_dim, super_embed_dim):
    return dropout * 1.0 * sample_embed_dim / super_embed_dim


class DropPath(nn.Module):
    """Per-sample stochastic depth (model/utils.py:68-98)."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.drop_prob or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * (mask / keep)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, dropout=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, pre_norm=True, scale=False,
                 relative_position=False, change_qkv=False, max_relative_position=14):
        super().__init__()
        self.super_embed_dim = dim
        self.super_mlp_ratio = mlp_ratio
        self.super_ffn_embed_dim_this_layer = int(mlp_ratio * dim)
        self.super_num_heads = num_heads
        self.normalize_before = pre_norm
        self.super_dropout = attn_drop
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.scale = scale
        self.relative_position = relative_position

        self.sample_embed_dim = None
        self.sample_mlp_ratio = None
        self.sample_ffn_embed_dim_this_layer = None
        self.sample_num_heads_this_layer = None
        self.sample_scale = None
        self.sample_dropout = None
        self.sample_attn_dropout = None
        self.is_identity_layer = None
        self.fused = True           # allow the fused-block execution under bf16 autocast
        self._dp = None             # drop-path scales handed down by the model for this forward

        self.attn = AttentionSuper(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                   attn_drop=attn_drop, proj_drop=dropout, scale=self.scale,
                                   relative_position=self.relative_position, change_qkv=change_qkv,
                                   max_relative_position=max_relative_position)
        self.attn_layer_norm = LayerNormSuper(self.super_embed_dim)
        self.ffn_layer_norm = LayerNormSuper(self.super_embed_dim)
        self.activation_fn = gelu
        self.fc1 = LinearSuper(super_in_dim=self.super_embed_dim, super_out_dim=self.super_ffn_embed_dim_this_layer)
        self.fc2 = LinearSuper(super_in_dim=self.super_ffn_embed_dim_this_layer, super_out_dim=self.super_embed_dim)

    def set_sample_config(self, is_identity_layer, sample_embed_dim=None, sample_mlp_ratio=None,
                          sample_num_heads=None, sample_dropout=None, sample_attn_dropout=None,
                          sample_out_dim=None):
        d = self.__dict__                      # plain attributes: skip nn.Module.__setattr__ (hot path of every step)
        if is_identity_layer:
            d['is_identity_layer'] = True
            return
        d['is_identity_layer'] = False
        d['sample_embed_dim'] = sample_embed_dim
        d['sample_out_dim'] = sample_out_dim
        d['sample_mlp_ratio'] = sample_mlp_ratio
        d['sample_ffn_embed_dim_this_layer'] = int(sample_embed_dim * sample_mlp_ratio)
        d['sample_num_heads_this_layer'] = sample_num_heads
        d['sample_dropout'] = sample_dropout
        d['sample_attn_dropout'] = sample_attn_dropout
        self.attn_layer_norm.set_sample_config(sample_embed_dim=sample_embed_dim)
        # head dim fixed at 64: supernet_transformer.py:243
        self.attn.set_sample_config(sample_q_embed_dim=sample_num_heads * 64, sample_num_heads=sample_num_heads,
                                    sample_in_embed_dim=sample_embed_dim)
        self.fc1.set_sample_config(sample_in_dim=sample_embed_dim, sample_out_dim=self.sample_ffn_embed_dim_this_layer)
        self.fc2.set_sample_config(sample_in_dim=self.sample_ffn_embed_dim_this_layer, sample_out_dim=sample_out_dim)
        self.ffn_layer_norm.set_sample_config(sample_embed_dim=sample_embed_dim)

    def maybe_layer_norm(self, layer_norm, x, before=False, after=False):
        assert before ^ after
        return layer_norm(x) if (after ^ self.normalize_before) else x

    def drop_path_scales(self, batch, device):
        """Per-sample scales mask/keep of the two DropPath applications of this block (or None)."""
        p = getattr(self.drop_path, 'drop_prob', None)
        if not p or not self.training:
            return None, None
        keep = 1.0 - p
        s = torch.floor(keep + torch.rand(2, batch, device=device)) / keep
        return s[0], s[1]

    def forward(self, x):
        if self.is_identity_layer:
            return x
        # bf16 throughput mode: the whole block is one autograd node on the HIP kernels
        if self.fused and _bf16_autocast(x) and _block.supported(self, x):
            dp = self._dp if self._dp is not None else self.drop_path_scales(x.shape[0], x.device)
            self._dp = None
            return _block.BlockFunction.apply(x, dp[0], dp[1], self)
        residual = x
        x = self.maybe_layer_norm(self.attn_layer_norm, x, before=True)
        x = self.attn(x)
        x = F.dropout(x, p=self.sample_attn_dropout, training=self.training)
        x = residual + self.drop_path(x)
        x = self.maybe_layer_norm(self.attn_layer_norm, x, after=True)

        residual = x
        x = self.maybe_layer_norm(self.ffn_layer_norm, x, before=True)
        x = self.activation_fn(self.fc1(x))
        x = F.dropout(x, p=self.sample_dropout, training=self.training)
        x = self.fc2(x)
        x = F.dropout(x, p=self.sample_dropout, training=self.training)
        if self.scale:
            x = x * (self.super_mlp_ratio / self.sample_mlp_ratio)
        x = residual + self.drop_path(x)
        x = self.maybe_layer_norm(self.ffn_layer_norm, x, after=True)
        return x

    def get_complexity(self, sequence_length):
        if self.is_identity_layer:
            return 0
        n = sequence_length + 1
        return (self.attn_layer_norm.get_complexity(n) + self.attn.get_complexity(n) +
                self.ffn_layer_norm.get_complexity(n) + self.fc1.get_complexity(n) + self.fc2.get_complexity(n))


NATIVE_ENDS = os.environ.get('CREAM_NATIVE_ENDS', '1') != '0'     # stem / tail of forward_features on csrc/stem_tail.hip (bf16 mode)


class Vision_TransformerSuper(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., pre_norm=True, scale=False, gp=False, relative_position=False,
                 change_qkv=False, abs_pos=True, max_relative_position=14):
        super().__init__()
        self.super_embed_dim = embed_dim
        self.super_mlp_ratio = mlp_ratio
        self.super_layer_num = depth
        self.super_num_heads = num_heads
        self.super_dropout = drop_rate
        self.super_attn_dropout = attn_drop_rate
        self.num_classes = num_classes
        self.pre_norm = pre_norm
        self.scale = scale
        self.patch_embed_super = PatchembedSuper(img_size=img_size, patch_size=patch_size,
                                                 in_chans=in_chans, embed_dim=embed_dim)
        self.gp = gp

        self.sample_embed_dim = None
        self.sample_mlp_ratio = None
        self.sample_layer_num = None
        self.sample_num_heads = None
        self.sample_dropout = None
        self.sample_output_dim = None

        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            TransformerEncoderLayer(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                    qk_scale=qk_scale, dropout=drop_rate, attn_drop=attn_drop_rate,
                                    drop_path=dpr[i], pre_norm=pre_norm, scale=self.scale,
                                    change_qkv=change_qkv, relative_position=relative_position,
                                    max_relative_position=max_relative_position)
            for i in range(depth)])

        num_patches = self.patch_embed_super.num_patches
        self.abs_pos = abs_pos
        if self.abs_pos:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
            _trunc_normal_(self.pos_embed, std=.02)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        _trunc_normal_(self.cls_token, std=.02)
        if self.pre_norm:
            self.norm = LayerNormSuper(super_embed_dim=embed_dim)
        self.head = LinearSuper(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self.apply(self._init_weights)

    def _init_weights(self, m):
        # supernet_transformer.py:82-89 dispatches on nn.Linear / nn.LayerNorm (our
        # Linear/LayerNorm supers subclass them for exactly this reason)
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'rel_pos_embed'}

    def get_classifier(self):
        return self.head

    def set_sample_config(self, config: dict):
        self.sample_embed_dim = config['embed_dim']
        self.sample_mlp_ratio = config['mlp_ratio']
        self.sample_layer_num = config['layer_num']
        self.sample_num_heads = config['num_heads']
        self.sample_dropout = calc_dropout(self.super_dropout, self.sample_embed_dim[0], self.super_embed_dim)
        self.patch_embed_super.set_sample_config(self.sample_embed_dim[0])
        self.sample_output_dim = list(self.sample_embed_dim[1:]) + [self.sample_embed_dim[-1]]
        for i, blk in enumerate(self.blocks):
            if i < self.sample_layer_num:
                blk.set_sample_config(
                    is_identity_layer=False,
                    sample_embed_dim=self.sample_embed_dim[i],
                    sample_mlp_ratio=self.sample_mlp_ratio[i],
                    sample_num_heads=self.sample_num_heads[i],
                    sample_dropout=calc_dropout(self.super_dropout, self.sample_embed_dim[i], self.super_embed_dim),
                    sample_out_dim=self.sample_output_dim[i],
                    sample_attn_dropout=calc_dropout(self.super_attn_dropout, self.sample_embed_dim[i],
                                                     self.super_embed_dim))
            else:
                blk.set_sample_config(is_identity_layer=True)
        if self.pre_norm:
            self.norm.set_sample_config(self.sample_embed_dim[-1])
        self.head.set_sample_config(self.sample_embed_dim[-1], self.num_classes)

    def get_sampled_params_numel(self, config):
        self.set_sample_config(config)
        total = 0
        for name, module in self.named_modules():
            if hasattr(module, 'calc_sampled_param_num'):
                parts = name.split('.')
                if parts[0] == 'blocks' and int(parts[1]) >= config['layer_num']:
                    continue
                total += module.calc_sampled_param_num()
        # cls token + position embedding of the sampled width (supernet_transformer.py:138)
        return total + self.sample_embed_dim[0] * (2 + self.patch_embed_super.num_patches)

    def get_complexity(self, sequence_length):
        flops = self.patch_embed_super.get_complexity(sequence_length)
        flops += np.prod(self.pos_embed[..., :self.sample_embed_dim[0]].size()) / 2.0
        for blk in self.blocks:
            flops += blk.get_complexity(sequence_length + 1)
        flops += self.head.get_complexity(sequence_length + 1)
        return flops

    def _keep_prob(self, probs, device):
        """(L, 1, 1) keep probabilities of the active blocks' DropPath, cached per configuration."""
        key = (probs, str(device))
        cache = self.__dict__.setdefault('_keep_cache', {})
        if key not in cache:
            cache[key] = 1.0 - torch.tensor(probs, device=device).view(-1, 1, 1)
        return cache[key]

    def forward_features(self, x):
        B = x.shape[0]
        E = self.sample_embed_dim[0]
        if x.is_cuda and NATIVE_ENDS and _block.stem_supported(self, x):
            # bf16 throughput mode: unfold + GEMM + class token + position embedding in three launches
            x = _block.stem(self, x)
        else:
            x = self.patch_embed_super(x)
            cls = self.cls_token[..., :E].expand(B, -1, -1)
            x = torch.cat((cls, x), dim=1)        # promotes like the reference (fp32 stream under autocast)
            if self.abs_pos:
                x = x + self.pos_embed[..., :E]
            x = F.dropout(x, p=self.sample_dropout, training=self.training)
        active = [blk for blk in self.blocks if not blk.is_identity_layer]
        if active and _bf16_autocast(x) and all(b.fused and _block.supported(b, x) for b in active):
            # bf16 throughput mode: the whole run of blocks is ONE autograd node on the HIP kernels
            scales = None
            probs = [(getattr(b.drop_path, 'drop_prob', 0.0) or 0.0) if self.training else 0.0 for b in active]
            if any(probs):
                # all drop-path draws of this forward in three launches instead of 3 per block
                keep = self._keep_prob(tuple(probs), x.device)
                scales = torch.floor(keep + torch.rand(len(active), 2, B, device=x.device)) / keep
            if (NATIVE_ENDS and _block.NATIVE_BLOCK and self.pre_norm and self.gp and x.shape[1] > 1
                    and self.norm.weight.dtype == torch.float32 and self.norm.bias is not None
                    and _block._no_frozen_parameters(self.norm)):
                # ... and the final LayerNorm + token mean ride on the same node (the last block's output stays pending)
                return _block.StackFunction.apply(x, scales, active, self.norm.weight, self.norm.bias, self.norm.eps)
            x = _block.StackFunction.apply(x, scales, active, None, None, 1e-5)
        else:
            for blk in self.blocks:
                x = blk(x)
        if self.pre_norm:
            x = self.norm(x)
        if self.gp:
            return torch.mean(x[:, 1:], dim=1)
        return x[:, 0]

    def forward(self, x):
        feat = self.forward_features(x)
        if feat.is_cuda and NATIVE_ENDS and not isinstance(self.head, nn.Identity) and _block.head_supported(self.head, feat):
            return _block.head(self.head, feat)          # bf16 throughput mode: classifier on the own GEMMs
        return self.head(feat)
