"""TEST INFRASTRUCTURE — CPU (PyTorch, fp32) restatement of the reference's AutoFormer
supernet forward, written functionally over a state_dict.

It follows the reference's DENSE formulation line by line (the (N, N, d) relative position
embeddings, the dense RPE bmm's), i.e. it is deliberately NOT the bucketed algorithm the
product uses — so that agreement between the two is evidence, not tautology.

Pinned against tests/golden/autoformer_*.npz, which were produced by importing and running
the reference's own Vision_TransformerSuper (tests/golden/make_golden.py).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Each function cites the reference lines it restates (paths relative to AutoFormer/).
"""
import math
import random

import torch
import torch.nn.functional as F


def sample_configs(choices):
    """supernet_engine.py:13-24 — draw order matters (CPython `random`): depth, then
    mlp_ratio per layer, then num_heads per layer, then ONE embed_dim for all layers."""
    config = {}
    depth = random.choice(choices['depth'])
    for dimension in ['mlp_ratio', 'num_heads']:
        config[dimension] = [random.choice(choices[dimension]) for _ in range(depth)]
    config['embed_dim'] = [random.choice(choices['embed_dim'])] * depth
    config['layer_num'] = depth
    return config


def rel_pos_embeddings(table_v, table_h, length, max_rel):
    """model/module/multihead_super.py:40-66 (RelativePosition2D_super.forward)."""
    n = length - 1
    side = int(n ** 0.5)
    rq = torch.arange(n)
    rk = torch.arange(n)
    dv = rk[None, :] // side - rq[:, None] // side
    dh = rk[None, :] % side - rq[:, None] % side
    fv = torch.clamp(dv, -max_rel, max_rel) + max_rel + 1
    fh = torch.clamp(dh, -max_rel, max_rel) + max_rel + 1
    fv = F.pad(fv, (1, 0, 1, 0), "constant", 0).long()
    fh = F.pad(fh, (1, 0, 1, 0), "constant", 0).long()
    return table_v[fv] + table_h[fh], fv, fh


def attention(sd, prefix, x, E, H, max_rel=14, change_qkv=True, relative_position=True, fc_scale=False,
              super_embed_dim=None):
    """model/module/multihead_super.py:133-160 with the sampling of qkv_super.py:72-83 and
    Linear_super.py:71-81."""
    B, N, C = x.shape
    Q = H * 64 if change_qkv else super_embed_dim
    w = sd[prefix + 'qkv.weight'][:, :E]
    if change_qkv:
        w = torch.cat([w[i:3 * Q:3, :] for i in range(3)], dim=0)      # qkv_super.py:72-77
    else:
        w = w[:3 * Q, :]
    b = sd.get(prefix + 'qkv.bias')
    b = b[:3 * Q] if b is not None else None                            # qkv_super.py:80-83
    scale = (Q // H) ** -0.5 if change_qkv else (E // H) ** -0.5        # multihead_super.py:102-110
    qkv = F.linear(x, w, b).reshape(B, N, 3, H, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    d = q.shape[-1]
    attn = (q @ k.transpose(-2, -1)) * scale
    if relative_position:
        r_p_k, _, _ = rel_pos_embeddings(sd[prefix + 'rel_pos_embed_k.embeddings_table_v'][:, :d],
                                         sd[prefix + 'rel_pos_embed_k.embeddings_table_h'][:, :d], N, max_rel)
        attn = attn + (q.permute(2, 0, 1, 3).reshape(N, H * B, -1) @ r_p_k.transpose(2, 1)) \
            .transpose(1, 0).reshape(B, H, N, N) * scale
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, -1)
    if relative_position:
        r_p_v, _, _ = rel_pos_embeddings(sd[prefix + 'rel_pos_embed_v.embeddings_table_v'][:, :d],
                                         sd[prefix + 'rel_pos_embed_v.embeddings_table_h'][:, :d], N, max_rel)
        attn_1 = attn.permute(2, 0, 1, 3).reshape(N, B * H, -1)
        out = out + (attn_1 @ r_p_v).transpose(1, 0).reshape(B, H, N, -1).transpose(2, 1).reshape(B, N, -1)
    if fc_scale:
        out = out * (super_embed_dim / Q)
    return F.linear(out, sd[prefix + 'proj.weight'][:E, :Q], sd[prefix + 'proj.bias'][:E])


def attention_core(qkv, tkv, tkh, tvv, tvh, scale, max_rel=14, attn_keep=None):
    """The part of model/module/multihead_super.py:135-154 between the qkv and proj GEMMs, in the
    reference's DENSE formulation: qkv (B, N, 3, H, d) -> (B, N, H, d).  (What the fused HIP
    kernels cream_attn_rpe2d_fwd/bwd replace; autograd of this function is their backward oracle.)"""
    B, N, _, H, d = qkv.shape
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
    attn = (q @ k.transpose(-2, -1)) * scale                                              # :138
    r_p_k, _, _ = rel_pos_embeddings(tkv, tkh, N, max_rel)                                # :140
    attn = attn + (q.permute(2, 0, 1, 3).reshape(N, H * B, -1) @ r_p_k.transpose(2, 1)) \
        .transpose(1, 0).reshape(B, H, N, N) * scale                                      # :141-142
    attn = attn.softmax(dim=-1)                                                           # :144
    if attn_keep is not None:               # :145 `attn = self.attn_drop(attn)` under a GIVEN mask: attn_keep (B, H, N, N) holds
        attn = attn * attn_keep             # keep / (1 - p) — nn.Dropout's scaling with the random draw taken out
    out = (attn @ v).transpose(1, 2).reshape(B, N, -1)                                    # :147
    r_p_v, _, _ = rel_pos_embeddings(tvv, tvh, N, max_rel)                                # :149
    attn_1 = attn.permute(2, 0, 1, 3).reshape(N, B * H, -1)
    out = out + (attn_1 @ r_p_v).transpose(1, 0).reshape(B, H, N, -1).transpose(2, 1).reshape(B, N, -1)   # :150-154
    return out.reshape(B, N, H, d)


def block(sd, i, x, E, H, ratio, **kw):
    """model/supernet_transformer.py:251-287 (pre-norm, dropout 0, no drop-path)."""
    p = f'blocks.{i}.'
    Fdim = int(E * ratio)
    h = F.layer_norm(x, (E,), sd[p + 'attn_layer_norm.weight'][:E], sd[p + 'attn_layer_norm.bias'][:E], 1e-5)
    x = x + attention(sd, p + 'attn.', h, E, H, **kw)
    h = F.layer_norm(x, (E,), sd[p + 'ffn_layer_norm.weight'][:E], sd[p + 'ffn_layer_norm.bias'][:E], 1e-5)
    h = F.gelu(F.linear(h, sd[p + 'fc1.weight'][:Fdim, :E], sd[p + 'fc1.bias'][:Fdim]).float())
    h = F.linear(h, sd[p + 'fc2.weight'][:E, :Fdim], sd[p + 'fc2.bias'][:E])
    return x + h


def forward(sd, config, images, patch=16, gp=True, **kw):
    """model/supernet_transformer.py:147-172 for a sampled config (embed_dim identical in
    every layer, as sample_configs produces)."""
    E = config['embed_dim'][0]
    B = images.shape[0]
    x = F.conv2d(images, sd['patch_embed_super.proj.weight'][:E], sd['patch_embed_super.proj.bias'][:E],
                 stride=patch).flatten(2).transpose(1, 2)                 # model/module/embedding_super.py:33-40
    cls = sd['cls_token'][..., :E].expand(B, -1, -1)
    x = torch.cat((cls, x), dim=1) + sd['pos_embed'][..., :E]
    for i in range(config['layer_num']):
        x = block(sd, i, x, E, config['num_heads'][i], config['mlp_ratio'][i], **kw)
    x = F.layer_norm(x, (E,), sd['norm.weight'][:E], sd['norm.bias'][:E], 1e-5)
    feat = torch.mean(x[:, 1:], dim=1) if gp else x[:, 0]
    return F.linear(feat, sd['head.weight'][:, :E], sd['head.bias'])


def soft_target_cross_entropy(logits, target):
    """timm.loss.SoftTargetCrossEntropy (not vendored; restated from its published
    definition): mean over the batch of sum(-target * log_softmax(logits))."""
    return torch.sum(-target * F.log_softmax(logits, dim=-1), dim=-1).mean()


def train_step(sd_params, config, images, target, **kw):
    """One forward/backward of the reference step body (supernet_engine.py:49-97) on CPU
    fp32; returns (loss, grads dict).  The optimizer (timm create_optimizer -> AdamW) is
    third-party code that is not vendored in the reference — parity unpinned there."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd_params.items()}
    loss = soft_target_cross_entropy(forward(params, config, images, **kw), target)
    loss.backward()
    return loss.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in params.items()}
