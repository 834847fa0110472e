"""iRPE — image relative position encoding (host-side mirror of
iRPE/DeiT-with-iRPE/irpe.py: bucket functions :18-257, bucket-id tables :260-415, the `iRPE`
module :418-693, `iRPE_Cross` :696-767, config builders :770-931), MI355X-first.

Same public names, constructor arguments, parameter names/shapes (`lookup_table_weight`,
`lookup_table_bias`) and numerical results as the reference; what differs is how it is
evaluated:

  * bucket ids depend only on the 2-D OFFSET between query and key positions, so the table is
    built once as a (2H-1) x (2W-1) offset map through a 1-D piecewise lookup (bit-exact with
    the reference's float32 formula, including half-to-even rounding and the float clip at
    beta) and expanded to the (L, L) int32 matrix the `rpe_index` operator takes — on the
    device, cached per (method, coefficients, H, W, skip, device);
  * contextual/transposed (rpe_q, rpe_k): one GEMM  x . W  (shared head: a single (B*H*L, d) x
    (d, nb) product, no transposes) followed by the HIP gather `cream_rpe_index_fwd`
    (backward: the deterministic scatter `cream_rpe_index_bwd`);
  * contextual/non-transposed (rpe_v): the reference gathers an (H, L, L, d) weight tensor
    (85 MB at L=577) and runs a batched matmul against it; here the probabilities are first
    summed per bucket with the scatter kernel, S[b,h,i,u] = sum_{j: idx[i,j]=u} P[b,h,i,j],
    then multiplied with the (nb, d) table — same sums, no (L, L, d) tensor.
"""
import math

import torch
import torch.nn as nn

from .autoformer.attention_op import rpe_gather, rpe_scatter


class METHOD:
    """irpe.py:117-127."""
    EUCLIDEAN = 0
    QUANT = 1
    PRODUCT = 3
    CROSS = 4
    CROSS_ROWS = 41
    CROSS_COLS = 42


class RPEConfig(dict):
    """Attribute-style dict (the reference uses easydict.EasyDict)."""
    __setattr__ = dict.__setitem__

    def __getattr__(self, name):
        if name.startswith("__"):          # copy / pickle probe special methods through getattr: they must see AttributeError
            raise AttributeError(name)
        return self.get(name)


@torch.no_grad()
def piecewise_index(relative_position, alpha, beta, gamma, dtype):
    """Eq. (18) of the iRPE paper as evaluated by irpe.py:18-52:
        idx = x                                                       |x| <= alpha
        idx = sign(x) * min(beta, round(alpha + ln(|x|/alpha)/ln(gamma/alpha) * (beta-alpha)))  otherwise
    in float32, `round` half-to-even, the clip against the FLOAT beta applied before the cast
    (truncation) to `dtype`.  Works for integer or float input of any shape."""
    x = relative_position
    xa = x.abs()
    far = xa > alpha
    xf = xa.to(torch.float32)
    safe = torch.where(far, xf, torch.full_like(xf, float(alpha) if alpha > 0 else 1.0))
    y = (alpha + torch.log(safe / alpha) / math.log(gamma / alpha) * (beta - alpha)).round().clip(max=beta)
    y = (torch.sign(x).to(torch.float32) * y).to(dtype)
    near = x.round().to(dtype) if x.dtype in (torch.float32, torch.float64) else x.to(dtype)
    return torch.where(far, y, near)


def get_num_buckets(method, alpha, beta, gamma):
    """irpe.py:260-283 (without the extra `skip` bucket)."""
    b = int(beta)
    return (2 * b + 1) ** 2 if method == METHOD.PRODUCT else 2 * b + 1


_OFFSET_CACHE = {}
_TABLE_CACHE = {}


@torch.no_grad()
def offset_bucket_map(method, height, width, alpha, beta, gamma):
    """(2H-1, 2W-1) int32 map: bucket id of the offset (dr, dc) = (query row - key row, query col -
    key col), stored at [dr + H - 1, dc + W - 1].  CPU, cached.  (irpe.py:130-257 + :348-349.)"""
    key = (method, height, width, float(alpha), float(beta), float(gamma))
    hit = _OFFSET_CACHE.get(key)
    if hit is not None:
        return hit
    dr = torch.arange(-(height - 1), height).view(-1, 1).expand(2 * height - 1, 2 * width - 1)
    dc = torch.arange(-(width - 1), width).view(1, -1).expand(2 * height - 1, 2 * width - 1)
    kw = dict(alpha=alpha, beta=beta, gamma=gamma, dtype=torch.long)
    b = int(beta)
    if method == METHOD.PRODUCT:
        ids = (piecewise_index(dr, **kw) + b) * (2 * b + 1) + (piecewise_index(dc, **kw) + b)
    elif method == METHOD.EUCLIDEAN:
        dis = (dr * dr + dc * dc).float().sqrt().round()
        ids = piecewise_index(dis, **kw) + b
    elif method == METHOD.QUANT:
        ids = piecewise_index(dr * dr + dc * dc, **kw) + b
    elif method == METHOD.CROSS_ROWS:
        ids = piecewise_index(dr, **kw) + b
    elif method == METHOD.CROSS_COLS:
        ids = piecewise_index(dc, **kw) + b
    else:
        raise NotImplementedError(f"[Error] The method ID {method} does not exist.")
    ids = ids.to(torch.int32).contiguous()
    _OFFSET_CACHE[key] = ids
    return ids


@torch.no_grad()
def get_bucket_ids_2d(method, height, width, skip, alpha, beta, gamma, dtype=torch.long,
                      device=torch.device('cpu')):
    """(skip + L, skip + L) bucket ids and the number of buckets including the extra bucket of
    the `skip` leading tokens (irpe.py:363-415).  Query i / key j at grid positions
    (i // W, i % W); rows and columns of the skip tokens hold the extra id."""
    device = torch.device(device)
    key = (method, height, width, skip, float(alpha), float(beta), float(gamma), dtype, str(device))
    hit = _TABLE_CACHE.get(key)
    if hit is not None:
        return hit
    omap = offset_bucket_map(method, height, width, alpha, beta, gamma).to(device)
    nb = get_num_buckets(method, alpha, beta, gamma)
    L = height * width
    pos = torch.arange(L, device=device)
    r, c = pos // width, pos % width
    ids = omap[(r[:, None] - r[None, :]) + height - 1, (c[:, None] - c[None, :]) + width - 1]
    if skip > 0:
        full = torch.full((skip + L, skip + L), nb, dtype=ids.dtype, device=device)
        full[skip:, skip:] = ids
        ids, nb = full, nb + 1
    out = (ids.to(dtype).contiguous(), nb)
    _TABLE_CACHE[key] = out
    return out


class iRPE(nn.Module):
    """irpe.py:418-693.  `mode` 'bias' | 'contextual'; transposed=True for rpe_q / rpe_k (the
    result is added to the (B, H, L, L) logits), False for rpe_v (the result is added to the
    (B, H, L, d) attention output)."""

    def __init__(self, head_dim, num_heads=8, mode=None, method=None, transposed=True, num_buckets=None,
                 initializer=None, rpe_config=None):
        super().__init__()
        assert mode in (None, 'bias', 'contextual')
        assert method is not None, 'method should be a METHOD ID rather than None'
        self.num_heads, self.head_dim = num_heads, head_dim
        self.mode, self.method, self.transposed, self.num_buckets = mode, method, transposed, num_buckets
        self.initializer = initializer if initializer is not None else (lambda x: None)
        self.rpe_config = rpe_config
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        if self.mode == 'bias':
            if not self.transposed:
                raise NotImplementedError("[Error] Bias non-transposed RPE does not exist.")
            self.lookup_table_bias = nn.Parameter(torch.zeros(self.num_heads, self.num_buckets))
            self.initializer(self.lookup_table_bias)
        elif self.mode == 'contextual':
            shape = (self.num_heads, self.head_dim, self.num_buckets) if self.transposed else \
                (self.num_heads, self.num_buckets, self.head_dim)
            self.lookup_table_weight = nn.Parameter(torch.zeros(shape))
            self.initializer(self.lookup_table_weight)

    def bucket_ids(self, x, height=None, width=None):
        """int32 (L, L) bucket matrix for this input (irpe.py:523-583; always int32 here — the
        rpe_index operator is always present)."""
        return self.bucket_ids_for(x.shape[2], x.device, height, width)

    def bucket_ids_for(self, L, device, height=None, width=None):
        if height is None:
            height = width = int(math.sqrt(L))
        skip = L - height * width
        c = self.rpe_config
        ids, nb = get_bucket_ids_2d(self.method, height, width, skip, c.alpha, c.beta, c.gamma,
                                    dtype=torch.int32, device=device)
        assert nb == self.num_buckets
        return ids

    def forward(self, x, height=None, width=None):
        idx = self.bucket_ids(x, height, width)
        if self.transposed:
            return self.forward_rpe_transpose(x, idx)
        return self.forward_rpe_no_transpose(x, idx)

    def forward_rpe_transpose(self, x, rp_bucket):
        """x (B, H, L, d) -> (B, H, L, L) [contextual]  or (1, H', L, L) [bias]  (irpe.py:585-647)."""
        Lq, Lk = rp_bucket.shape
        if self.mode == 'bias':
            return self.lookup_table_bias[:, rp_bucket.flatten().long()].view(1, self.num_heads, Lq, Lk)
        w = self.lookup_table_weight                                     # (H', d, nb)
        mm = _matmul_for(x, w)
        lookup = mm(x, w[0]) if w.shape[0] == 1 else mm(x, w.unsqueeze(0))
        return rpe_gather(lookup, rp_bucket)

    def forward_rpe_no_transpose(self, x, rp_bucket):
        """x = attention probabilities (B, H, L, L) -> (B, H, L, d)  (irpe.py:649-687)."""
        assert self.mode == 'contextual', "Only support contextual version in non-transposed version"
        w = self.lookup_table_weight                                     # (H', nb, d)
        s = rpe_scatter(x, rp_bucket, self.num_buckets).to(w.dtype)      # (B, H, L, nb)
        mm = _matmul_for(s, w)
        return mm(s, w[0]) if w.shape[0] == 1 else mm(s, w.unsqueeze(0))

    def __repr__(self):
        return ('iRPE(head_dim={r.head_dim}, num_heads={r.num_heads}, mode="{r.mode}", method={r.method}, '
                'transposed={r.transposed}, num_buckets={r.num_buckets}, rpe_config={r.rpe_config})').format(r=self)


def _matmul_for(x, w):
    """The lookup products of irpe.py:641-644 / :683-687: fp32 device tensors outside autocast multiply on the own
    exact-fp32 kernel (cream_bmm_f32, 4-D inputs), everything else on the framework's matmul."""
    from .autoformer import native_fp32
    return native_fp32.matmul if x.dim() == 4 and native_fp32.usable(x, w) else torch.matmul


class iRPE_Cross(nn.Module):
    """irpe.py:696-767: rows + cols, two iRPE modules."""

    def __init__(self, method, **kwargs):
        super().__init__()
        assert method == METHOD.CROSS
        self.rp_rows = iRPE(**kwargs, method=METHOD.CROSS_ROWS)
        self.rp_cols = iRPE(**kwargs, method=METHOD.CROSS_COLS)

    def forward(self, x, height=None, width=None):
        return self.rp_rows(x, height=height, width=width) + self.rp_cols(x, height=height, width=width)

    # -- one-table view for the fused kernels (csrc/irpe_attn.hip) --------------------------------------------
    # rows + cols is ONE lookup per (i, j) into a table over the (row bucket, column bucket) PAIRS that occur:
    #   x W_r[:, b_r(i,j)] + x W_c[:, b_c(i,j)] = x (W_r[:, b_r] + W_c[:, b_c])        (same for bias / value tables)
    # With the zoo's ratio (1.9, skip 1) the pairs are the product method's 50 buckets (7 x 7 + the class-token one).
    mode = property(lambda self: self.rp_rows.mode)
    transposed = property(lambda self: self.rp_rows.transposed)
    method = METHOD.CROSS

    @torch.no_grad()
    def merged_ids_for(self, L, device, height=None, width=None):
        """-> (int32 (L, L) ids of the occurring (row, col) bucket pairs, long (nb,) row bucket of each pair,
        long (nb,) column bucket of each pair, nb); cached per (L, device, grid)."""
        key = (L, str(device), height, width)
        cache = self.__dict__.setdefault("_merged", {})
        hit = cache.get(key)
        if hit is None:
            br = self.rp_rows.bucket_ids_for(L, device, height, width).long()
            bc = self.rp_cols.bucket_ids_for(L, device, height, width).long()
            pairs, inv = torch.unique(br * self.rp_cols.num_buckets + bc, return_inverse=True)
            hit = (inv.to(torch.int32).contiguous(), pairs // self.rp_cols.num_buckets,
                   pairs % self.rp_cols.num_buckets, int(pairs.numel()))
            cache[key] = hit
        return hit

    def merged_table(self, L, device, height=None, width=None):
        """The pair table as a differentiable function of the two parameters: (H', nb) bias, (H', d, nb) transposed
        contextual, (H', nb, d) value side.  fp32, contiguous."""
        _, ir, ic, _ = self.merged_ids_for(L, device, height, width)
        if self.mode == 'bias':
            r, c, dim = self.rp_rows.lookup_table_bias, self.rp_cols.lookup_table_bias, 1
        else:
            r, c, dim = self.rp_rows.lookup_table_weight, self.rp_cols.lookup_table_weight, 2 if self.transposed else 1
        return (r.index_select(dim, ir) + c.index_select(dim, ic)).contiguous()


def get_single_rpe_config(ratio=1.9, method=METHOD.PRODUCT, mode='contextual', shared_head=True, skip=0):
    """irpe.py:770-819: alpha = ratio, beta = 2 ratio, gamma = 8 ratio; +1 bucket if skip > 0."""
    c = RPEConfig(shared_head=shared_head, mode=mode, method=method,
                  alpha=1 * ratio, beta=2 * ratio, gamma=8 * ratio)
    c.num_buckets = get_num_buckets(method, c.alpha, c.beta, c.gamma) + (1 if skip > 0 else 0)
    return c


def get_rpe_config(ratio=1.9, method=METHOD.PRODUCT, mode='contextual', shared_head=True, skip=0, rpe_on='k'):
    """irpe.py:822-893."""
    if isinstance(method, str):
        method = dict(euc=METHOD.EUCLIDEAN, quant=METHOD.QUANT, cross=METHOD.CROSS,
                      product=METHOD.PRODUCT)[method.lower()]
    if mode == 'ctx':
        mode = 'contextual'
    kw = dict(ratio=ratio, method=method, mode=mode, shared_head=shared_head, skip=skip)
    return RPEConfig(rpe_q=get_single_rpe_config(**kw) if 'q' in rpe_on else None,
                     rpe_k=get_single_rpe_config(**kw) if 'k' in rpe_on else None,
                     rpe_v=get_single_rpe_config(**kw) if 'v' in rpe_on else None)


def build_rpe(config, head_dim, num_heads):
    """irpe.py:896-931 -> [rpe_q, rpe_k, rpe_v]; q and k are transposed, v is not."""
    if config is None:
        return None, None, None

    def one(rpe, transposed):
        if rpe is None:
            return None
        cls = iRPE if rpe.method != METHOD.CROSS else iRPE_Cross
        return cls(head_dim=head_dim, num_heads=1 if rpe.shared_head else num_heads, mode=rpe.mode,
                   method=rpe.method, transposed=transposed, num_buckets=rpe.num_buckets, rpe_config=rpe)
    return [one(config.rpe_q, True), one(config.rpe_k, True), one(config.rpe_v, False)]
