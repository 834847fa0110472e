"""Per-step gradient averaging across the GPUs of one node — the data-parallel exchange of
the supernet step (reference: implicit in torch DistributedDataParallel,
AutoFormer/supernet_train.py:286-289, `find_unused_parameters=True`).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm; "gloo"
for the CPU tests).  Design points, MI355X-first rather than DDP-reducer-first:

  * gradients live in flat fp32 bucket buffers from the start (`p.grad` are views), one
    bucket per transformer block plus stem and tail — a bucket is a single large,
    contiguous RCCL message (supernet-S: ~10 MB per block) and its completion is a
    per-layer event, not a 25 MB size threshold;
  * a bucket's all-reduce is launched from a post-accumulate-grad hook the moment its last
    gradient has been written, on a dedicated side stream that waits on an event of the
    compute stream — communication of block i overlaps backward of blocks i-1..0;
  * every rank samples the SAME sub-network (random.seed(epoch), supernet_engine.py:36),
    so the set of parameters that receive gradients is known up front: blocks beyond the
    sampled depth are not "unused parameters" to be discovered, their buckets are simply
    not sent (their gradients stay exactly zero on every rank).  No dead bytes on xGMI.
  * ACTIVE-SLICE messages: a sampled sub-network only writes W.grad[:out, :in] of every super weight, and the
    configuration is the same on every rank — so a bucket's message is the contiguous gather of those slices
    (one HIP launch to pack, one to scatter the result back: csrc/slices.hip), not the full super bucket:
    supernet-S sends 52-139 MB per step depending on the sampled widths instead of always 139 MB;
  * averaging = one AVG all-reduce where the backend has it (RCCL), else SUM followed by an
    in-place 1/world scale (gloo) — DDP semantics; lr is already scaled by the world size,
    supernet_train.py:294;
  * like DistributedDataParallel, construction BROADCASTS rank 0's parameters and buffers: the
    reference seeds `torch.manual_seed(args.seed + rank)` (supernet_train.py:197-198), so without it
    every rank would start from different weights and gradient averaging alone never reconciles them.
"""
import weakref

import torch
import torch.distributed as dist


def attention_layout(model):
    """{block index: (change_qkv, super_embed_dim)} read off the modules: with `change_qkv=False`
    (multihead_super.py:81-84, 104-106) the qkv projection is a LinearSuper whose sampled output is ALWAYS
    3 * super_embed_dim and the output projection reads super_embed_dim columns — the head count does not slice them."""
    out = {}
    for i, blk in enumerate(getattr(model, "blocks", [])):
        attn = getattr(blk, "attn", None)
        if attn is not None and hasattr(attn, "change_qkv"):
            out[i] = (bool(attn.change_qkv), int(getattr(attn, "super_embed_dim", 0)))
    return out


def autoformer_active_slice(name, p, config, layout=None):
    """(rows, cols) of the part of parameter `name` (viewed as (numel / last dim, last dim); the patch-embedding
    convolution as (out, C*ph*pw)) that a sub-network with `config` can write — the slicing rules of
    Linear_super.py:71-81, qkv_super.py:72-83 (rows 3 i + j, i < Q = rows [0, 3 Q)), layernorm_super.py:26-37,
    embedding_super.py:33-40 and supernet_transformer.py:147-172.  `layout` = attention_layout(model): blocks built
    with `change_qkv=False` keep the whole qkv output / projection input (multihead_super.py:104-106); without a
    layout the attention projections are sent whole (always correct).  Unknown names: the whole tensor."""
    E = config["embed_dim"][0]
    last = p.shape[-1] if p.dim() > 1 else p.numel()
    full = (p.numel() // last, last)
    parts = name.split(".")
    if parts[0] == "blocks":
        i = int(parts[1])
        if i >= config["layer_num"]:
            return (0, 0)
        F_ = int(E * config["mlp_ratio"][i])
        leaf = ".".join(parts[2:])
        change_qkv, super_e = (layout or {}).get(i, (None, 0))
        if change_qkv is None:                                # layout unknown: never drop a written element
            if leaf in ("attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight"):
                return full
            Q = 0
        else:
            Q = 64 * config["num_heads"][i] if change_qkv else super_e
        rule = {"attn.qkv.weight": (3 * Q, E), "attn.qkv.bias": (1, 3 * Q), "attn.proj.weight": (E, Q),
                "attn.proj.bias": (1, E), "fc1.weight": (F_, E), "fc1.bias": (1, F_), "fc2.weight": (E, F_),
                "fc2.bias": (1, E), "attn_layer_norm.weight": (1, E), "attn_layer_norm.bias": (1, E),
                "ffn_layer_norm.weight": (1, E), "ffn_layer_norm.bias": (1, E)}.get(leaf)
        return rule if rule is not None else full
    if name == "patch_embed_super.proj.weight":
        return (E, p.numel() // p.shape[0])
    if name in ("patch_embed_super.proj.bias", "norm.weight", "norm.bias", "cls_token"):
        return (1, E)
    if name in ("pos_embed", "head.weight"):
        return (full[0], E)
    return full


class GradReducer:
    """mode 'allreduce' (default): one all-reduce per bucket.  mode 'rs_ag': reduce-scatter + all-gather
    per bucket — on the xGMI full mesh (7 point-to-point links per GPU) each phase moves 1/world of the
    bucket over EVERY link at once instead of the whole bucket around a ring (2 x 17 MB per
    link per step for supernet-S instead of ~244 MB over one); buckets are padded to a multiple of the
    world size inside the arena.  Same API, same results (fixed reduction order per shard)."""

    def __init__(self, model, process_group=None, bucket_of=None, world=None, mode="allreduce", slice_of=autoformer_active_slice):
        assert mode in ("allreduce", "rs_ag")
        self.mode = mode
        self.slice_of = slice_of              # None: always send whole buckets
        # slice_of(name, param, config, layout) -> (rows, cols) written by the step; a callable of the earlier three-argument
        # form (name, param, config) keeps working
        self._slice_nargs = 4
        if slice_of is not None:
            try:
                import inspect
                params = inspect.signature(slice_of).parameters.values()
                if not any(q.kind == q.VAR_POSITIONAL for q in params):
                    self._slice_nargs = min(4, sum(q.kind in (q.POSITIONAL_ONLY, q.POSITIONAL_OR_KEYWORD) for q in params))
            except (TypeError, ValueError):
                pass
        self.layout = attention_layout(model)  # change_qkv / super_embed_dim per block, from the modules themselves
        self.stage = None                     # staging arena of the packed messages (allocated on first use)
        self._slice_tables = {}
        self.msg = {}
        self.model = model
        self.pg = process_group
        # `world` overrides the group size (tests drive the bucket logic with a stubbed collective)
        self.world = world if world is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self.params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.device = self.params[0][1].device
        self.on_gpu = self.device.type == "cuda"
        bucket_of = bucket_of or self._default_bucket_of
        names = []
        for n, _ in self.params:
            b = bucket_of(n)
            if b not in names:
                names.append(b)
        self.bucket_names = names
        self.members = {b: [] for b in names}
        for n, p in self.params:
            self.members[bucket_of(n)].append((n, p))
        # flat buffers; p.grad become views (zero-filled).  All buckets are slices of ONE allocation
        # (bucket starts 64-element aligned): zero_grad is a single memset
        self.flat = {}
        sizes = {b: sum(p.numel() for _, p in mem) for b, mem in self.members.items()}
        starts, total = {}, 0
        quantum = 64 * max(1, self.world)                         # equal, 256-byte aligned shards for rs_ag
        self.padded = {}
        for b in names:
            starts[b] = total
            self.padded[b] = (sizes[b] + quantum - 1) // quantum * quantum
            total += self.padded[b]
        self.arena = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.starts, self.quantum = starts, quantum
        self.flat_padded = {}
        for b, mem in self.members.items():
            buf = self.arena[starts[b]:starts[b] + sizes[b]]
            self.flat_padded[b] = self.arena[starts[b]:starts[b] + self.padded[b]]
            off = 0
            for _, p in mem:
                p.grad = buf[off:off + p.numel()].view_as(p)
                off += p.numel()
            self.flat[b] = buf
        self.bucket_index = {id(p): b for b, mem in self.members.items() for _, p in mem}
        self.pending = {}
        self._seen = set()
        self.works = []
        self.active = set()
        self.bytes_sent = 0
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.use_avg = False
        if self.world > 1:
            if dist.is_initialized():
                # what DDP does at wrap time (supernet_train.py:286-289): same weights on every rank
                with torch.no_grad():
                    for t in list(model.parameters()) + list(model.buffers()):
                        dist.broadcast(t.data, src=0, group=self.pg)
                # the broadcast wrote through .data (no version bump): bf16 operand copies made by a forward that
                # ran BEFORE the reducer existed would stay on the pre-broadcast weights of ranks != 0
                from .autoformer import block as _blk
                _blk.refresh_operands(model, force=True)
                self.use_avg = dist.get_backend(self.pg) == "nccl"
            # hooks hold the reducer WEAKLY: a discarded reducer must not keep firing (or stay alive)
            ref = weakref.ref(self)

            def hook(p, ref=ref):
                me = ref()
                if me is not None:
                    me._hook(p)

            for _, p in self.params:
                p.register_post_accumulate_grad_hook(hook)
            # fused blocks write their parameter gradients themselves and announce them
            from .autoformer import block as _block

            def ready(params, ref=ref):
                me = ref()
                if me is None:
                    _block.remove_grads_ready(ready)
                else:
                    for p in params:
                        me._hook(p)

            self._ready_cb = _block.on_grads_ready(ready)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        cb = getattr(self, "_ready_cb", None)
        if cb is not None:
            from .autoformer import block as _block
            _block.remove_grads_ready(cb)
            self._ready_cb = None

    @staticmethod
    def _default_bucket_of(name):
        parts = name.split(".")
        if parts[0] == "blocks":
            return f"block{int(parts[1]):02d}"
        if parts[0] in ("norm", "head"):
            return "tail"
        return "stem"

    # ------------------------------------------------------------------ per step
    def zero_grad(self):
        """Zero-fill (not None, see engine.SupernetTrainer) — one memset for all buckets."""
        self.arena.zero_()

    def owns_grads(self):
        """True while every parameter's .grad is still the view into the arena handed out at construction (an
        optimizer.zero_grad(set_to_none=True) or a `p.grad = ...` assignment breaks the aliasing silently)."""
        lo, hi = self.arena.data_ptr(), self.arena.data_ptr() + self.arena.numel() * 4
        return all(p.grad is not None and lo <= p.grad.data_ptr() < hi for _, p in self.params)

    def prepare(self, config=None):
        """Call after zero_grad and before forward.  `config` is the sampled architecture
        (same on all ranks); blocks >= layer_num will not produce gradients."""
        self.works = []
        self.bytes_sent = 0
        depth = config["layer_num"] if config is not None else None
        self.active = set()
        self.pending = {}
        self._seen = set()
        for b, mem in self.members.items():
            if depth is not None and b.startswith("block") and int(b[5:]) >= depth:
                continue
            self.active.add(b)
            self.pending[b] = len(mem)
        self.msg = {}
        if config is not None and self.slice_of is not None and self.world > 1:
            for b in self.active:
                self.msg[b] = self._slice_plan(b, config)

    # ------------------------------------------------------------------ active-slice messages
    def _slice_plan(self, b, config):
        """-> (message view into the staging arena, pack / unpack plan) of bucket b for this configuration; cached per
        (bucket, slice signature) — the S search space has 27 signatures per block."""
        if self._slice_nargs >= 4:
            sig = tuple(self.slice_of(n, p, config, self.layout) for n, p in self.members[b])
        else:
            sig = tuple(self.slice_of(n, p, config) for n, p in self.members[b])
        key = (b, sig)
        plan = self._slice_tables.get(key)
        if plan is None:
            if self.stage is None:
                self.stage = torch.zeros_like(self.arena)
            off, views, jobs, max_rows = 0, [], [], 0
            for (n, p), (rows, cols) in zip(self.members[b], sig):
                if rows * cols == 0:
                    continue
                last = p.shape[-1] if p.dim() > 1 else p.numel()
                if n == "patch_embed_super.proj.weight":
                    last = p.numel() // p.shape[0]
                g2d = p.grad.view(-1, last)
                assert rows <= g2d.shape[0] and cols <= last, (n, rows, cols, tuple(p.shape))
                views.append((g2d, rows, cols, off))
                jobs.append((g2d.data_ptr(), last, off, rows, cols))
                max_rows = max(max_rows, rows)
                off += rows * cols
            padded = (off + self.quantum - 1) // self.quantum * self.quantum
            msg = self.stage[self.starts[b]:self.starts[b] + (padded if self.mode == "rs_ag" else off)]
            table = None
            if self.on_gpu:
                from . import _lib
                arr = (_lib.SliceJob * len(jobs))()
                base = self.stage.data_ptr() + self.starts[b] * 4
                for j, (ptr, ld, o, r, c) in zip(arr, jobs):
                    j.full, j.ld, j.packed_off, j.rows, j.cols = ptr, ld, o, r, c
                table = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device), len(jobs), max_rows, base)
            plan = self._slice_tables[key] = (msg, views, table, off)
        return plan

    def _copy_slices(self, plan, to_packed):
        msg, views, table, _ = plan
        if table is not None:
            from . import _lib
            tab, n, max_rows, base = table
            _lib.check(_lib.load().cream_slices_copy(tab.data_ptr(), n, base, max_rows, 1 if to_packed else 0,
                                                     torch.cuda.current_stream(self.device).cuda_stream), "cream_slices_copy")
            return
        for g2d, rows, cols, off in views:               # host tensors (gloo tests): plain views
            m = msg[off:off + rows * cols].view(rows, cols)
            if to_packed:
                m.copy_(g2d[:rows, :cols])
            else:
                g2d[:rows, :cols].copy_(m)

    def _hook(self, p):
        b = self.bucket_index.get(id(p))
        if b is None or b not in self.pending or id(p) in self._seen:
            return
        # a parameter counts ONCE per step: nodes that write their gradients themselves announce them explicitly
        # (block.on_grads_ready), and autograd may still run the parameter's accumulate hook with an undefined gradient
        self._seen.add(id(p))
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def _reduce_scatter_gather(self, buf):
        """In-place average of `buf` (numel % world == 0) as reduce-scatter + all-gather."""
        w = self.world
        shard = buf.numel() // w
        rank = dist.get_rank(self.pg)
        mine = buf[rank * shard:(rank + 1) * shard]
        if dist.get_backend(self.pg) == "nccl":
            dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.AVG, group=self.pg)
        else:
            # backends without reduce_scatter (gloo): exchange the shards, add them in rank order
            recv = torch.empty_like(buf)
            dist.all_to_all_single(recv, buf.clone(), group=self.pg)
            acc = recv[:shard].clone()
            for r in range(1, w):
                acc += recv[r * shard:(r + 1) * shard]
            mine.copy_(acc / w)
        dist.all_gather_into_tensor(buf, mine.clone(), group=self.pg)

    def _launch(self, b):
        plan = self.msg.get(b)
        buf = plan[0] if plan is not None else (self.flat_padded[b] if self.mode == "rs_ag" else self.flat[b])
        self.bytes_sent += (plan[3] if plan is not None else self.flat[b].numel()) * 4
        if self.on_gpu:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                if plan is not None:
                    self._copy_slices(plan, True)
                if self.mode == "rs_ag":
                    self._reduce_scatter_gather(buf)
                elif self.use_avg:
                    dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.pg)      # one pass over the message
                else:
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
                    buf.mul_(1.0 / self.world)
                if plan is not None:
                    self._copy_slices(plan, False)
            return
        if plan is not None:
            self._copy_slices(plan, True)
        if self.mode == "rs_ag":
            self._reduce_scatter_gather(buf)
            if plan is not None:
                self._copy_slices(plan, False)
            return
        w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self.works.append((w, buf, plan))

    def finish(self):
        """Call after backward, before the optimizer step."""
        if self.world == 1:
            return
        # buckets whose hooks did not all fire (should not happen) are flushed here so that
        # ranks can never diverge silently
        for b, left in list(self.pending.items()):
            if left > 0:
                self.pending[b] = 0
                self._launch(b)
        if self.on_gpu:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        else:
            for w, buf, plan in self.works:
                w.wait()
                buf.mul_(1.0 / self.world)
                if plan is not None:
                    self._copy_slices(plan, False)
        self.works = []


# CU budget with world > 1.  Every hot kernel of the library is a persistent grid of one or two workgroups per CU holding most of
# the CU's LDS, so an RCCL kernel launched beside them (one workgroup per channel) gets a CU only when a grid drains.  The
# data-parallel driver therefore reserves R CUs (cream_cu_reserve: every persistent grid is sized for CUs - R) and caps RCCL at
# R channels.  R = 16 by default: 2 CUs per XCD, 6 % of the chip; the per-step message is 52-139 MB in ~27 buckets, i.e. a few
# MB per bucket over 7 xGMI links x ~50 GB/s effective — 16 channels are more than the links need (unmeasured on hardware: gpurun
# boxes have one GPU; CREAM_COMM_CUS overrides, 0 = no reserve / RCCL defaults).  bench.py records the values in config.comm.
DEFAULT_COMM_CUS = 16
_comm_budget = {"reserved_cus": 0, "rccl_channels": None, "cus_for_kernels": None}


def comm_cu_budget():
    """What init_distributed reserved: {'reserved_cus', 'rccl_channels', 'cus_for_kernels'}."""
    return dict(_comm_budget)


def reserve_comm_cus(world, backend):
    import os
    r = int(os.environ.get("CREAM_COMM_CUS", DEFAULT_COMM_CUS if (world > 1 and backend == "nccl") else 0))
    if r > 0 and backend == "nccl":
        # must be in the environment before RCCL initialises (first collective of the process group)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(r))
        os.environ.setdefault("NCCL_MIN_NCHANNELS", str(min(r, 4)))
        _comm_budget["rccl_channels"] = {"max": int(os.environ["NCCL_MAX_NCHANNELS"]), "min": int(os.environ["NCCL_MIN_NCHANNELS"])}
    if torch.cuda.is_available():
        from . import _lib
        lib = _lib.load()
        lib.cream_cu_reserve(max(r, 0))
        _comm_budget["cus_for_kernels"] = lib.cream_cu_count()
    _comm_budget["reserved_cus"] = max(r, 0)
    return r


def default_comm_mode(world):
    """'allreduce' (one collective per bucket) or 'rs_ag' (reduce-scatter + all-gather: every xGMI link carries 1 / world of a
    bucket in each phase).  Selectable at run time: CREAM_COMM_MODE=allreduce|rs_ag (bench.py --comm-mode auto reads it).  The
    default is 'allreduce' at every world size: a step moves 52-139 MB in ~9 ms, i.e. < 30 GB/s per GPU against 7 links of
    ~50 GB/s — the exchange is launch-count-bound, not link-bound, and rs_ag doubles the collectives per bucket (no multi-GPU
    measurement exists to overrule this; both modes are bit-identical in the gloo tests)."""
    import os
    m = os.environ.get("CREAM_COMM_MODE")
    return m if m in ("allreduce", "rs_ag") else "allreduce"


def init_distributed(backend=None):
    """torch.distributed bootstrap from the torchrun environment (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*), as AutoFormer/lib/utils.py:209-235 does for the reference.
    Returns (rank, local_rank, world)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # CREAM_DIST_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks
            backend = os.environ.get("CREAM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC between the ranks of a node (see bench.py)
            torch.cuda.set_device(local)
        reserve_comm_cus(world, backend)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        dist.barrier()
    return rank, local, world
