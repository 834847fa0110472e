"""DETR's encoder self-attention with 2-D image relative position encoding — host-side mirror of
iRPE/DETR-with-iRPE/models/rpe_attention/multi_head_attention.py:17-193 (`RPEMultiheadAttention`) and
rpe_attention_function.py:20-389 (`rpe_multi_head_attention_forward`), on cream_amd.irpe (HIP `rpe_index` gather /
scatter underneath).  Call site in the reference: models/transformer.py:49-69 (rpe_config from `--enc_rpe2d`, skip = 0)
and its encoder layer, which passes `hw=(height, width)` of the backbone feature map — rectangular in general — and a
key padding mask for the padded images of a batch.

Sequence-first tensors (L, N, E) like `nn.MultiheadAttention`, packed `in_proj_weight` / `in_proj_bias`, `out_proj`,
`rpe_q / rpe_k / rpe_v` with the reference's parameter names: DETR-with-iRPE checkpoints load.  Semantics kept:
    q = (x_q W_q + b_q) * head_dim^-0.5 ;  A = q k^T + rpe_k(q) + rpe_q(k * head_dim^-0.5)^T
    A[attn_mask] / A[:, :, :, key_padding_mask] = -inf ;  P = dropout(softmax(A)) ;  out = P v + rpe_v(P) ;  out_proj
and, with `need_weights`, the head-averaged P is returned as well (:383-387).

DETR's heads are 32 wide (256 / 8): the fused iRPE attention kernels (head_dim 64) do not apply, the map is formed.
What `nn.MultiheadAttention` offers beyond DETR's use (`add_bias_kv`, `add_zero_attn`, `kdim` / `vdim`, static k / v)
is refused at construction instead of being silently approximated.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .irpe import build_rpe


class RPEMultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False, kdim=None,
                 vdim=None, rpe_config=None):
        super().__init__()
        if add_bias_kv or add_zero_attn or (kdim not in (None, embed_dim)) or (vdim not in (None, embed_dim)):
            raise NotImplementedError("RPEMultiheadAttention: add_bias_kv / add_zero_attn / kdim / vdim are outside "
                                      "DETR-with-iRPE's use of this module (models/transformer.py:142-176)")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        if bias:
            self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        else:
            self.register_parameter('in_proj_bias', None)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self._reset_parameters()
        self.rpe_q, self.rpe_k, self.rpe_v = build_rpe(rpe_config, head_dim=self.head_dim, num_heads=num_heads)

    def _reset_parameters(self):
        nn.init.xavier_uniform_(self.in_proj_weight)
        if self.in_proj_bias is not None:
            nn.init.constant_(self.in_proj_bias, 0.)
            nn.init.constant_(self.out_proj.bias, 0.)

    def _project(self, query, key, value):
        E, w, b = self.embed_dim, self.in_proj_weight, self.in_proj_bias
        if query is key and key is value:
            return F.linear(query, w, b).chunk(3, dim=-1)

        def part(x, i):
            return F.linear(x, w[i * E:(i + 1) * E], None if b is None else b[i * E:(i + 1) * E])

        return part(query, 0), part(key, 1), part(value, 2)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None, hw=None):
        """query (L, N, E), key / value (S, N, E); key_padding_mask (N, S) bool / byte; attn_mask (L, S) or
        (N * heads, L, S), bool or additive float; hw = (height, width) of the feature map when an rpe is present."""
        L, N, E = query.shape
        S = key.shape[0]
        H, hd = self.num_heads, self.head_dim
        assert E == self.embed_dim and key.shape[:2] == value.shape[:2]
        q, k, v = self._project(query, key, value)
        q = q * float(hd) ** -0.5                                                       # :235
        q = q.reshape(L, N, H, hd).permute(1, 2, 0, 3)                                    # (N, H, L, hd)
        k = k.reshape(S, N, H, hd).permute(1, 2, 0, 3)
        v = v.reshape(S, N, H, hd).permute(1, 2, 0, 3)
        attn = q @ k.transpose(-2, -1)                                                    # (N, H, L, S)
        if self.rpe_k is not None or self.rpe_q is not None or self.rpe_v is not None:
            assert hw is not None and L == S == hw[0] * hw[1], "image rpe needs hw = (height, width) with L = S = h * w"
        if self.rpe_k is not None:
            attn = attn + self.rpe_k(q, height=hw[0], width=hw[1])                        # :328-331
        if self.rpe_q is not None:
            attn = attn + self.rpe_q(k * float(hd) ** -0.5, height=hw[0], width=hw[1]).transpose(-2, -1)    # :334-338
        if attn_mask is not None:
            if attn_mask.dtype == torch.uint8:
                attn_mask = attn_mask.to(torch.bool)
            m = attn_mask if attn_mask.dim() == 2 else attn_mask.view(N, H, L, S)
            assert m.shape[-2:] == (L, S)
            attn = attn.masked_fill(m, float("-inf")) if m.dtype == torch.bool else attn + m          # :343-347
        if key_padding_mask is not None:
            assert key_padding_mask.shape == (N, S)
            attn = attn.masked_fill(key_padding_mask.to(torch.bool)[:, None, None, :], float("-inf"))   # :349-357
        attn = F.dropout(attn.softmax(dim=-1), p=self.dropout, training=self.training)
        out = attn @ v
        if self.rpe_v is not None:
            out = out + self.rpe_v(attn, height=hw[0], width=hw[1])                       # :371-377
        out = self.out_proj(out.permute(2, 0, 1, 3).reshape(L, N, E))
        return out, (attn.sum(dim=1) / H if need_weights else None)
