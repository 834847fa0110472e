"""Weight-entangled supernet operators of AutoFormer, MI355X-first.

Host-side mirror of the reference's `model/module/*` classes (same constructor
arguments, attributes, `set_sample_config` protocol, parameter names and state-dict keys,
so published `supernet-*.pth` checkpoints load and `model/supernet_transformer.py` can use
them unchanged — see cream_amd/dropin/model):

  LinearSuper              AutoFormer/model/module/Linear_super.py:6-81
  qkv_super                AutoFormer/model/module/qkv_super.py:7-83
  LayerNormSuper           AutoFormer/model/module/layernorm_super.py:5-45
  PatchembedSuper          AutoFormer/model/module/embedding_super.py:7-48
  RelativePosition2D_super AutoFormer/model/module/multihead_super.py:14-66
  AttentionSuper           AutoFormer/model/module/multihead_super.py:68-160

What differs is HOW the sampled sub-network is evaluated:
  * the active `W[:out, :in]` block is read in place from the super weight (leading
    dimension = super dim) — no per-step weight materialisation except the small
    row-regrouping of qkv (3Q x E elements);
  * the relative-position bias never exists as an (N, N, d) tensor: attention runs through
    cream_amd.autoformer.attention_op, i.e. the bucketed identity of SURVEY Appendix B.1
    (`q.T_k^T` lookups + rpe_index gather, rpe_index scatter + `S.T_v`) executed by the
    HIP kernels of libcream_amd.so;
  * patch embedding is a GEMM over unfolded patches instead of a strided convolution.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import attention_op, native_fp32


def _trunc_normal_(t, std):
    # same distribution as the reference's model/utils.py:trunc_normal_ (a=-2, b=2)
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


class _LazySamples(dict):
    """`module.samples` of the reference ({'weight': W[:out, :in] view, 'bias': ...}), filled on
    first access after a `set_sample_config`: the supernet step calls set_sample_config on ~90
    modules per step, and the fused execution never looks at these views."""

    def __init__(self, fill):
        super().__init__()
        self._fill = fill
        self.stale = True

    def _ready(self):
        if self.stale:
            self.stale = False
            self._fill()

    def __getitem__(self, k):
        self._ready()
        return dict.__getitem__(self, k)

    def __contains__(self, k):
        self._ready()
        return dict.__contains__(self, k)

    def get(self, k, default=None):
        self._ready()
        return dict.get(self, k, default)

    def keys(self):
        self._ready()
        return dict.keys(self)

    def items(self):
        self._ready()
        return dict.items(self)


class _SampledLinearBase(nn.Linear):
    """Shared bookkeeping of LinearSuper / qkv_super: a super (largest) weight plus the
    extents of the currently sampled sub-matrix."""

    def __init__(self, super_in_dim, super_out_dim, bias=True, scale=False):
        super().__init__(super_in_dim, super_out_dim, bias=bias)
        self.super_in_dim = super_in_dim
        self.super_out_dim = super_out_dim
        self.sample_in_dim = None
        self.sample_out_dim = None
        self.samples = _LazySamples(self._fill_samples)
        self.scale = scale
        self.profiling = False

    def profile(self, mode=True):
        self.profiling = mode

    def sample_parameters(self, resample=False):
        if self.profiling or resample:
            return self._sample_parameters()
        return self.samples

    def set_sample_config(self, sample_in_dim, sample_out_dim):
        d = self.__dict__                      # plain attributes: skip nn.Module.__setattr__ (hot: ~90 calls per step)
        d['sample_in_dim'] = sample_in_dim
        d['sample_out_dim'] = sample_out_dim
        d['sample_scale'] = self.super_out_dim / sample_out_dim
        self.samples.stale = True              # Linear_super.py:45-49 slices here; the views are built on first use

    def _slice_weight(self):
        raise NotImplementedError

    def _fill_samples(self):
        dict.__setitem__(self.samples, 'weight', self._slice_weight())
        dict.__setitem__(self.samples, 'bias', self.bias[:self.sample_out_dim] if self.bias is not None else None)

    def _sample_parameters(self):
        self.samples.stale = False
        self._fill_samples()
        self.sample_scale = self.super_out_dim / self.sample_out_dim
        return self.samples

    def forward(self, x):
        if x.is_cuda and native_fp32.usable(x, self.weight, self.bias):
            # fp32 parity mode on the device: the framework's own exact-fp32 matrix-core GEMMs on the super weight in place
            seg, step = self._native_row_map()
            y = native_fp32.linear(x, self.weight, self.bias, self.sample_out_dim, self.sample_in_dim, seg, step)
            return y * self.sample_scale if self.scale else y
        self.sample_parameters()
        y = F.linear(x, self.samples['weight'], self.samples['bias'])
        return y * self.sample_scale if self.scale else y

    def _native_row_map(self):
        return 0, 0

    def calc_sampled_param_num(self):
        assert 'weight' in self.samples
        n = self.samples['weight'].numel()
        if self.samples['bias'] is not None:
            n += self.samples['bias'].numel()
        return n

    def get_complexity(self, sequence_length):
        return sequence_length * np.prod(self.samples['weight'].size())


class LinearSuper(_SampledLinearBase):
    """Top-left block `W[:out, :in]`, `b[:out]` (Linear_super.py:71-81) as strided views."""

    def __init__(self, super_in_dim, super_out_dim, bias=True, uniform_=None, non_linear='linear', scale=False):
        super().__init__(super_in_dim, super_out_dim, bias=bias, scale=scale)
        # Linear_super.py:32-36 — xavier-uniform weight, zero bias
        if uniform_ is None:
            nn.init.xavier_uniform_(self.weight)
        else:
            uniform_(self.weight, non_linear=non_linear)
        if bias:
            nn.init.constant_(self.bias, 0.)

    def _slice_weight(self):
        return self.weight[:self.sample_out_dim, :self.sample_in_dim]


class qkv_super(_SampledLinearBase):
    """qkv projection whose OUTPUT rows are interleaved in the super weight
    (qkv_super.py:72-77): q = rows 0,3,6,.., k = rows 1,4,7,.., v = rows 2,5,8,.. of
    `W[:3Q, :E]`, while the bias is the plain prefix `b[:3Q]` (qkv_super.py:80-83) —
    q-bias = b[0:Q], k-bias = b[Q:2Q], v-bias = b[2Q:3Q].  Both quirks are reproduced."""

    def __init__(self, super_in_dim, super_out_dim, bias=True, uniform_=None, non_linear='linear', scale=False):
        super().__init__(super_in_dim, super_out_dim, bias=bias, scale=scale)
        # (the reference leaves nn.Linear's default init here: qkv_super.py:24)

    def _native_row_map(self):
        # output n of [q | k | v] = super row 3 (n % Q) + n / Q  (qkv_super.py:75)
        return self.sample_out_dim // 3, 3

    def _slice_weight(self):
        n3 = self.sample_out_dim
        assert n3 % 3 == 0, "qkv sample_out_dim must be a multiple of 3"
        w = self.weight[:n3, :self.sample_in_dim]
        # (Q, 3, E) -> (3, Q, E): rows regrouped as [q | k | v]; one small copy per step
        return w.reshape(n3 // 3, 3, self.sample_in_dim).transpose(0, 1).reshape(n3, self.sample_in_dim)


class LayerNormSuper(nn.LayerNorm):
    """LayerNorm over the first `sample_embed_dim` channels (layernorm_super.py:26-37)."""

    def __init__(self, super_embed_dim):
        super().__init__(super_embed_dim)
        self.super_embed_dim = super_embed_dim
        self.sample_embed_dim = None
        self.samples = _LazySamples(self._fill_samples)
        self.profiling = False

    def profile(self, mode=True):
        self.profiling = mode

    def sample_parameters(self, resample=False):
        if self.profiling or resample:
            return self._sample_parameters()
        return self.samples

    def _fill_samples(self):
        dict.__setitem__(self.samples, 'weight', self.weight[:self.sample_embed_dim])
        dict.__setitem__(self.samples, 'bias', self.bias[:self.sample_embed_dim])

    def _sample_parameters(self):
        self.samples.stale = False
        self._fill_samples()
        return self.samples

    def set_sample_config(self, sample_embed_dim):
        self.__dict__['sample_embed_dim'] = sample_embed_dim
        self.samples.stale = True

    def forward(self, x):
        if x.is_cuda and self.sample_embed_dim % 4 == 0 and native_fp32.usable(x, self.weight, self.bias):
            return native_fp32.layer_norm(x, self.weight, self.bias, self.sample_embed_dim, self.eps)
        self.sample_parameters()
        return F.layer_norm(x, (self.sample_embed_dim,), weight=self.samples['weight'],
                            bias=self.samples['bias'], eps=self.eps)

    def calc_sampled_param_num(self):
        return self.samples['weight'].numel() + self.samples['bias'].numel()

    def get_complexity(self, sequence_length):
        return sequence_length * self.sample_embed_dim


class PatchembedSuper(nn.Module):
    """Patch embedding with a sampled output width (embedding_super.py:27-40).  The
    parameter stays an `nn.Conv2d` named `proj` (checkpoint keys `patch_embed_super.proj.*`);
    the evaluation is `unfold -> (B*P, C*ph*pw) x W[:E].view(E, -1)^T`, the same sums as
    the stride=kernel convolution."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, scale=False):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.super_embed_dim = embed_dim
        self.scale = scale
        self.sample_embed_dim = None
        self.sampled_weight = None
        self.sampled_bias = None
        self.sampled_scale = None

    def set_sample_config(self, sample_embed_dim):
        self.sample_embed_dim = sample_embed_dim
        self.sampled_weight = self.proj.weight[:sample_embed_dim, ...]
        self.sampled_bias = self.proj.bias[:sample_embed_dim, ...]
        if self.scale:
            self.sampled_scale = self.super_embed_dim / sample_embed_dim

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        ph, pw = self.patch_size
        gh, gw = H // ph, W // pw
        if x.is_cuda:
            from . import block as _block
            if _block.patch_embed_supported(self, x):        # own GEMMs (csrc/gemm_mfma.hpp), bf16 autocast
                return _block.patch_embed(self, x)
        patches = x.reshape(B, C, gh, ph, gw, pw).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * ph * pw)
        if x.is_cuda and native_fp32.usable(x, self.proj.weight, self.proj.bias):
            y = native_fp32.linear(patches, self.proj.weight, self.proj.bias, self.sample_embed_dim, C * ph * pw)
        else:
            y = F.linear(patches, self.sampled_weight.reshape(self.sample_embed_dim, -1), self.sampled_bias)
        return y * self.sampled_scale if self.scale else y

    def calc_sampled_param_num(self):
        return self.sampled_weight.numel() + self.sampled_bias.numel()

    def get_complexity(self, sequence_length):
        flops = 0
        if self.sampled_bias is not None:
            flops += self.sampled_bias.size(0)
        return flops + sequence_length * np.prod(self.sampled_weight.size())


def relative_index_tables(length, max_relative_position, device=None):
    """Row/column bucket indices of multihead_super.py:40-62 for a (length x length)
    attention map whose token 0 is the class token: grid side = int(sqrt(length-1)),
    vertical = k//side - q//side, horizontal = k%side - q%side, clamped to
    +-max_relative_position, shifted by max+1; row/column 0 (class token) -> bucket 0.
    Returns two contiguous int32 (length, length) tensors (iv, ih)."""
    n = length - 1
    side = int(n ** 0.5)
    pos = torch.arange(n, device=device)
    dv = (pos[None, :] // side - pos[:, None] // side).clamp(-max_relative_position, max_relative_position)
    dh = (pos[None, :] % side - pos[:, None] % side).clamp(-max_relative_position, max_relative_position)
    iv = torch.zeros(length, length, dtype=torch.int32, device=device)
    ih = torch.zeros(length, length, dtype=torch.int32, device=device)
    iv[1:, 1:] = (dv + max_relative_position + 1).to(torch.int32)
    ih[1:, 1:] = (dh + max_relative_position + 1).to(torch.int32)
    return iv.contiguous(), ih.contiguous()


class RelativePosition2D_super(nn.Module):
    """Two learnable (2*max+2, head_dim) tables, vertical and horizontal
    (multihead_super.py:16-38).  `forward(length_q, length_k)` still returns the dense
    (N, N, d) embedding for API compatibility, but AttentionSuper never calls it: it uses
    `tables()` and `index_tables()` and the bucketed kernels instead."""

    def __init__(self, num_units, max_relative_position):
        super().__init__()
        self.num_units = num_units
        self.max_relative_position = max_relative_position
        self.embeddings_table_v = nn.Parameter(torch.randn(max_relative_position * 2 + 2, num_units))
        self.embeddings_table_h = nn.Parameter(torch.randn(max_relative_position * 2 + 2, num_units))
        _trunc_normal_(self.embeddings_table_v, std=.02)
        _trunc_normal_(self.embeddings_table_h, std=.02)
        self.sample_head_dim = None          # (the sampled tables are properties below)
        self._index_cache = {}

    def set_sample_config(self, sample_head_dim):
        self.__dict__['sample_head_dim'] = sample_head_dim

    # multihead_super.py:32-35 slices the tables in set_sample_config; here the views are properties
    @property
    def sample_embeddings_table_h(self):
        return None if self.sample_head_dim is None else self.embeddings_table_h[:, :self.sample_head_dim]

    @property
    def sample_embeddings_table_v(self):
        return None if self.sample_head_dim is None else self.embeddings_table_v[:, :self.sample_head_dim]

    def calc_sampled_param_num(self):
        return self.sample_embeddings_table_h.numel() + self.sample_embeddings_table_v.numel()

    @property
    def num_buckets(self):
        return 2 * self.max_relative_position + 2

    def index_tables(self, length, device=None):
        device = device if device is not None else self.embeddings_table_v.device
        key = (length, str(device))
        hit = self._index_cache.get(key)
        if hit is None:
            hit = relative_index_tables(length, self.max_relative_position, device)
            self._index_cache[key] = hit
        return hit

    def tables(self):
        return self.sample_embeddings_table_v, self.sample_embeddings_table_h

    def forward(self, length_q, length_k):
        assert length_q == length_k, "square attention maps only (as in the reference's callers)"
        iv, ih = self.index_tables(length_q)
        return self.sample_embeddings_table_v[iv.long()] + self.sample_embeddings_table_h[ih.long()]


class AttentionSuper(nn.Module):
    """Multi-head self-attention with sampled width / head count and 2-D relative position
    bias on keys and values (multihead_super.py:68-160).

    forward(x: (B, N, E)) -> (B, N, E):
        qkv -> (q k^T + rpe_k(q)) * scale -> softmax -> dropout -> attn v + rpe_v(attn) -> proj
    where the scale multiplies BOTH the content and the position logits (:138-142), and
    rpe_v consumes the post-dropout attention (:145-154).  The core between the qkv and
    proj GEMMs runs in cream_amd.autoformer.attention_op (HIP kernels, no (N,N,d) tensors).
    """

    def __init__(self, super_embed_dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.,
                 normalization=False, relative_position=False, num_patches=None, max_relative_position=14,
                 scale=False, change_qkv=False):
        super().__init__()
        self.num_heads = num_heads
        head_dim = super_embed_dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.super_embed_dim = super_embed_dim
        self.fc_scale = scale
        self.change_qkv = change_qkv
        if change_qkv:
            self.qkv = qkv_super(super_embed_dim, 3 * super_embed_dim, bias=qkv_bias)
        else:
            self.qkv = LinearSuper(super_embed_dim, 3 * super_embed_dim, bias=qkv_bias)
        self.relative_position = relative_position
        if self.relative_position:
            self.rel_pos_embed_k = RelativePosition2D_super(super_embed_dim // num_heads, max_relative_position)
            self.rel_pos_embed_v = RelativePosition2D_super(super_embed_dim // num_heads, max_relative_position)
        self.max_relative_position = max_relative_position
        self.sample_qk_embed_dim = None
        self.sample_v_embed_dim = None
        self.sample_num_heads = None
        self.sample_scale = None
        self.sample_in_embed_dim = None
        self.proj = LinearSuper(super_embed_dim, super_embed_dim)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        # 'auto' | 'fused' | 'bucketed' — which HIP execution of the attention core to use
        self.attention_impl = 'auto'

    def set_sample_config(self, sample_q_embed_dim=None, sample_num_heads=None, sample_in_embed_dim=None):
        d = self.__dict__                      # plain attributes (see _SampledLinearBase.set_sample_config)
        d['sample_in_embed_dim'] = sample_in_embed_dim
        d['sample_num_heads'] = sample_num_heads
        if not self.change_qkv:
            d['sample_qk_embed_dim'] = self.super_embed_dim
            d['sample_scale'] = (sample_in_embed_dim // sample_num_heads) ** -0.5
        else:
            d['sample_qk_embed_dim'] = sample_q_embed_dim
            d['sample_scale'] = (sample_q_embed_dim // sample_num_heads) ** -0.5
        self.qkv.set_sample_config(sample_in_dim=sample_in_embed_dim, sample_out_dim=3 * self.sample_qk_embed_dim)
        self.proj.set_sample_config(sample_in_dim=self.sample_qk_embed_dim, sample_out_dim=sample_in_embed_dim)
        if self.relative_position:
            self.rel_pos_embed_k.set_sample_config(self.sample_qk_embed_dim // sample_num_heads)
            self.rel_pos_embed_v.set_sample_config(self.sample_qk_embed_dim // sample_num_heads)

    def calc_sampled_param_num(self):
        return 0

    def get_complexity(self, sequence_length):
        flops = self.qkv.get_complexity(sequence_length)
        flops += 2 * sequence_length * sequence_length * self.sample_qk_embed_dim
        flops += self.proj.get_complexity(sequence_length)
        if self.relative_position:
            flops += self.max_relative_position * sequence_length * sequence_length + sequence_length * sequence_length / 2.0
            flops += self.max_relative_position * sequence_length * sequence_length + sequence_length * self.sample_qk_embed_dim / 2.0
        return flops

    def forward(self, x):
        B, N, C = x.shape
        H = self.sample_num_heads
        qkv = self.qkv(x).reshape(B, N, 3, H, -1)
        drop_p = self.attn_drop.p if self.training else 0.0
        if self.relative_position:
            tkv, tkh = self.rel_pos_embed_k.tables()
            tvv, tvh = self.rel_pos_embed_v.tables()
            iv, ih = self.rel_pos_embed_k.index_tables(N, x.device)
            out = attention_op.attention_rpe2d(qkv, tkv, tkh, tvv, tvh, iv, ih, self.sample_scale,
                                               dropout_p=drop_p, impl=self.attention_impl,
                                               max_relative_position=self.max_relative_position)
        else:
            out = attention_op.attention_plain(qkv, self.sample_scale, dropout_p=drop_p)
        out = out.reshape(B, N, -1)
        if self.fc_scale:
            out = out * (self.super_embed_dim / self.sample_qk_embed_dim)
        out = self.proj(out)
        return self.proj_drop(out)
