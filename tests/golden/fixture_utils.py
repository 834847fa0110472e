"""Deterministic parameter / input fills shared by make_golden.py (which runs the
reference) and by the tests (which run the oracle and the product): same torch CPU
generator, same seeds, same order => identical tensors on both sides without shipping
the weights."""
import zlib

import torch

# supernet search spaces of AutoFormer/experiments/supernet/supernet-{T,S,B}.yaml
SUPERNETS = {
    'T': dict(embed_dim=256, depth=14, num_heads=4, mlp_ratio=4.0,
              choices=dict(mlp_ratio=[3.5, 4], num_heads=[3, 4], depth=[12, 13, 14], embed_dim=[192, 216, 240])),
    'S': dict(embed_dim=448, depth=14, num_heads=7, mlp_ratio=4.0,
              choices=dict(mlp_ratio=[3.0, 3.5, 4.0], num_heads=[5, 6, 7], depth=[12, 13, 14], embed_dim=[320, 384, 448])),
    'B': dict(embed_dim=640, depth=16, num_heads=10, mlp_ratio=4.0,
              choices=dict(mlp_ratio=[3.0, 3.5, 4.0], num_heads=[8, 9, 10], depth=[14, 15, 16], embed_dim=[528, 576, 624])),
}

# published subnets of AutoFormer/experiments/subnet/AutoFormer-{T,S}.yaml (RETRAIN sections)
SUBNET_T = dict(layer_num=13, embed_dim=[192] * 13,
                mlp_ratio=[3.5, 3.5, 3.0, 3.5, 3.0, 3.0, 4.0, 4.0, 3.5, 4.0, 3.5, 4.0, 3.5],
                num_heads=[3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 4, 3, 3])
SUBNET_S = dict(layer_num=13, embed_dim=[384] * 13,
                mlp_ratio=[3.0, 3.5, 3.0, 3.5, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 4.0, 3.5, 4.0],
                num_heads=[6, 6, 5, 7, 5, 5, 5, 6, 6, 7, 7, 6, 7])


def model_kwargs(size, drop_path_rate=0.0):
    s = SUPERNETS[size]
    # constructor call of AutoFormer/supernet_train.py:255-265 with the README flags
    return dict(img_size=224, patch_size=16, embed_dim=s['embed_dim'], depth=s['depth'], num_heads=s['num_heads'],
                mlp_ratio=s['mlp_ratio'], qkv_bias=True, drop_rate=0.0, drop_path_rate=drop_path_rate, gp=True,
                num_classes=1000, max_relative_position=14, relative_position=True, change_qkv=True, abs_pos=True)


def fill_params(model_or_state, seed=0):
    """Overwrite every parameter with seeded values that make each term of the model
    matter (position tables and biases are NOT near zero as after init)."""
    items = model_or_state.named_parameters() if hasattr(model_or_state, 'named_parameters') \
        else model_or_state.items()
    with torch.no_grad():
        for name, p in items:
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 7919)) & 0x7fffffff)
            if name.endswith('norm.weight') or name.endswith('layer_norm.weight'):
                v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            elif 'embeddings_table' in name:
                v = 0.5 * torch.randn(p.shape, generator=g)
            elif p.dim() <= 1 or name.endswith('.bias'):
                v = 0.05 * torch.randn(p.shape, generator=g)
            elif name in ('pos_embed', 'cls_token'):
                v = 0.2 * torch.randn(p.shape, generator=g)
            else:
                fan_in = p.shape[1] if p.dim() == 2 else p[0].numel()
                v = torch.randn(p.shape, generator=g) * (1.0 / fan_in ** 0.5)
            p.copy_(v)


def make_batch(batch, seed=0, img=224):
    g = torch.Generator().manual_seed(1000 + seed)
    images = torch.randn(batch, 3, img, img, generator=g)
    labels = torch.randint(0, 1000, (batch,), generator=g)
    # soft targets as Mixup would hand them over (label smoothing 0.1)
    target = torch.full((batch, 1000), 0.1 / 1000)
    target[torch.arange(batch), labels] += 0.9
    return images, target


def grad_digest(grads, stride=997):
    """Small fingerprint of a gradient dict: per-tensor L2 norm, sum, and a strided sample."""
    out = {}
    for k in sorted(grads):
        g = grads[k].detach().double().flatten()
        out[k + '|norm'] = g.norm().reshape(1)
        out[k + '|sum'] = g.sum().reshape(1)
        out[k + '|sample'] = g[::stride].clone()
    return out
