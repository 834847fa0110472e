"""Supernet step driver — host-side mirror of AutoFormer/supernet_engine.py:13-107 and of
the model / optimizer construction in AutoFormer/supernet_train.py:255-298.

Per step (supernet_engine.py:49-102): draw a random sub-network with CPython's `random`
(seeded per epoch with `random.seed(epoch)`, so every rank draws the SAME architecture),
`set_sample_config`, forward under autocast, soft-target cross entropy, zero_grad,
backward, optimizer step.  The reference trains under fp16 autocast + loss scaling;
here the compute dtype is bf16 (no scaler needed), or fp32 for parity runs.

timm (Mixup, SoftTargetCrossEntropy, create_optimizer, NativeScaler) is a third-party
dependency of the reference that is not vendored; the pieces used by the step are
restated from their published definitions (soft-target CE, AdamW with timm's
no-weight-decay rule) — their parity is unpinned, see DESIGN.md.
"""
import random

import torch
import torch.nn.functional as F

from . import block as _block
from .supernet import Vision_TransformerSuper

# AutoFormer/experiments/supernet/supernet-{T,S,B}.yaml
SEARCH_SPACES = {
    'T': dict(embed_dim=256, depth=14, num_heads=4, mlp_ratio=4.0,
              choices=dict(mlp_ratio=[3.5, 4], num_heads=[3, 4], depth=[12, 13, 14], embed_dim=[192, 216, 240])),
    'S': dict(embed_dim=448, depth=14, num_heads=7, mlp_ratio=4.0,
              choices=dict(mlp_ratio=[3.0, 3.5, 4.0], num_heads=[5, 6, 7], depth=[12, 13, 14],
                           embed_dim=[320, 384, 448])),
    'B': dict(embed_dim=640, depth=16, num_heads=10, mlp_ratio=4.0,
              choices=dict(mlp_ratio=[3.0, 3.5, 4.0], num_heads=[8, 9, 10], depth=[14, 15, 16],
                           embed_dim=[528, 576, 624])),
}


def sample_configs(choices):
    """supernet_engine.py:13-24.  Draw order: depth, mlp_ratio x depth, num_heads x depth,
    then ONE embed_dim shared by all layers."""
    depth = random.choice(choices['depth'])
    config = {dim: [random.choice(choices[dim]) for _ in range(depth)] for dim in ('mlp_ratio', 'num_heads')}
    config['embed_dim'] = [random.choice(choices['embed_dim'])] * depth
    config['layer_num'] = depth
    return config


def build_supernet(size='S', drop_path_rate=0.1, num_classes=1000, img_size=224, **overrides):
    """Vision_TransformerSuper as constructed by supernet_train.py:255-265 with the README
    recipe flags (--gp --change_qkv --relative_position, drop-path 0.1, drop 0.0)."""
    s = SEARCH_SPACES[size]
    kw = dict(img_size=img_size, patch_size=16, embed_dim=s['embed_dim'], depth=s['depth'],
              num_heads=s['num_heads'], mlp_ratio=s['mlp_ratio'], qkv_bias=True, drop_rate=0.0,
              drop_path_rate=drop_path_rate, gp=True, num_classes=num_classes, max_relative_position=14,
              relative_position=True, change_qkv=True, abs_pos=True)
    kw.update(overrides)
    return Vision_TransformerSuper(**kw)


def soft_target_cross_entropy(logits, target):
    """timm.loss.SoftTargetCrossEntropy: mean_b sum_c -t[b,c] log_softmax(x)[b,c] (fp32).  Device tensors: one launch
    that also produces the logit gradient (csrc/stem_tail.hip: cream_soft_ce)."""
    if logits.is_cuda and _block.soft_ce_supported(logits, target):
        return _block.SoftTargetCEFunction.apply(logits, target)
    return torch.sum(-target * F.log_softmax(logits.float(), dim=-1), dim=-1).mean()


def param_groups(model, weight_decay=0.05):
    """timm.optim.optim_factory.add_weight_decay as called by create_optimizer: 1-D tensors,
    biases and the names in model.no_weight_decay() get no decay.  (The reference's skip
    list contains 'rel_pos_embed', which matches no parameter name exactly, so the
    relative-position tables ARE decayed — reproduced.)"""
    skip = model.no_weight_decay() if hasattr(model, 'no_weight_decay') else set()
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.dim() <= 1 or name.endswith('.bias') or name in skip) else decay).append(p)
    return [{'params': no_decay, 'weight_decay': 0.0}, {'params': decay, 'weight_decay': weight_decay}]


class NativeAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias-corrected moments, no amsgrad) for a
    CUDA model as ONE kernel launch per step over every tensor (csrc/optim.hip), which also rewrites
    the bf16 operand copies (and their transposes) the block GEMMs read.  State layout = torch's
    (`exp_avg`, `exp_avg_sq`, `step` per parameter), so checkpoints move between the two.

    Two documented differences from torch.optim.AdamW: ONE global step counter (torch counts per parameter and skips
    parameters whose grad is None) — blocks beyond every sampled depth so far therefore see bias correction and weight
    decay from step 1 here, as they do under the reference's torch 1.7 once they were active;
    `load_state_dict` takes max(step) and creates zero moments for parameters the file has no state for.

    Every parameter must have a `.grad` tensor when `step()` runs (the trainer zero-fills instead of
    setting None — SURVEY 8e: under the reference's torch 1.7 every tensor that was active once keeps
    receiving weight decay and moment decay); the device-resident job table is rebuilt only if a
    gradient tensor was replaced (e.g. when a GradReducer re-homes them into its arena)."""

    def __init__(self, model, param_groups, lr, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(param_groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0))
        self.model = model
        self._cream_rewrites_operands = True     # (block._register: no invalidation needed after this optimizer's step)
        self._steps = 0
        self._table = None
        self._grads = ()
        self._plist = ()
        self._ops = []
        for g in self.param_groups:
            for p in g['params']:
                self.state[p] = dict(exp_avg=torch.zeros_like(p, memory_format=torch.contiguous_format),
                                     exp_avg_sq=torch.zeros_like(p, memory_format=torch.contiguous_format))

    def _build(self):
        wd = {p: g['weight_decay'] for g in self.param_groups for p in g['params']}
        for p in wd:                             # a torch.optim.AdamW checkpoint holds no state for never-active parameters
            st = self.state[p]
            for k in ('exp_avg', 'exp_avg_sq'):
                if k not in st or st[k].device != p.device:
                    st[k] = (torch.zeros_like(p, memory_format=torch.contiguous_format) if k not in st
                             else st[k].to(p.device).contiguous())
        states = {p: (st['exp_avg'], st['exp_avg_sq']) for p, st in self.state.items()}
        for p in wd:
            if p.grad is None:
                p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
        jobs, done, self._ops = [], set(), []
        for m in self.model.modules():
            if hasattr(m, 'attn') and hasattr(m, 'fc1') and hasattr(m, 'fc2'):
                ops = _block.operands(m, fresh=False)
                self._ops.append(ops)
                for p, j in ops.jobs(grads=True, states=states):
                    j.weight_decay = wd[p]
                    jobs.append(j)
                    done.add(p)
        for m in self.model.modules():
            if hasattr(m, 'proj') and hasattr(m, 'patch_size') and hasattr(m, 'sample_embed_dim'):     # PatchembedSuper
                ops = _block.patch_operands(m, fresh=False)
                self._ops.append(ops)
                for p, j in ops.jobs(grads=True, states=states):
                    if p in wd:
                        j.weight_decay = wd[p]
                        jobs.append(j)
                        done.add(p)
        head = getattr(self.model, 'head', None)
        if head is not None and hasattr(head, 'sample_in_dim') and getattr(head, 'bias', None) is not None and head.weight in wd:
            ops = _block.head_operands(head, fresh=False)
            self._ops.append(ops)
            for p, j in ops.jobs(grads=True, states=states):
                j.weight_decay = wd[p]
                jobs.append(j)
                done.add(p)
        for p in wd:
            if p not in done:
                jobs.append(_block.param_job(p.detach(), p.grad, states[p][0], states[p][1], weight_decay=wd[p]))
        self._table = _block.JobTable(jobs, next(iter(wd)).device)
        self._plist = tuple(wd)
        self._grads = tuple(p.grad for p in wd)
        self._ptrs = self._pointer_key()

    def _pointer_key(self):
        """Raw pointers baked into the device job table: model.to(), `p.data = ...`, re-created moments or an added
        parameter group move them without touching the gradient objects."""
        return tuple((p.data_ptr(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr())
                     if 'exp_avg' in self.state[p] else (p.data_ptr(), 0, 0)
                     for g in self.param_groups for p in g['params'])

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        if (self._table is None or any(p.grad is not g for p, g in zip(self._plist, self._grads))
                or self._ptrs != self._pointer_key()):
            self._build()
        g0 = self.param_groups[0]
        lr = g0['lr']
        assert all(g['lr'] == lr for g in self.param_groups), "NativeAdamW: one learning rate for all groups"
        self._steps += 1
        self._table.launch(update=True, lr=float(lr), beta1=g0['betas'][0], beta2=g0['betas'][1], eps=g0['eps'],
                           step=self._steps)
        for ops in self._ops:
            ops.mark_fresh()                 # the kernel rewrote every operand copy

    def state_dict(self):
        for st in self.state.values():
            st['step'] = torch.tensor(float(self._steps))
        return super().state_dict()

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        steps = [int(st['step']) for st in self.state.values() if 'step' in st]
        self._steps = max(steps) if steps else 0
        self._table = None                   # moments were re-created: new pointers


def build_optimizer(model, lr=5e-4, batch_size=128, world_size=1, weight_decay=0.05):
    """AdamW with the linear lr scaling of supernet_train.py:294: lr * batch * world / 512.  On the
    GPU the native single-launch optimizer; on the CPU (host-logic tests) torch.optim.AdamW."""
    scaled = lr * batch_size * world_size / 512.0
    groups = param_groups(model, weight_decay)
    if next(model.parameters()).is_cuda:
        return NativeAdamW(model, groups, lr=scaled, betas=(0.9, 0.999), eps=1e-8)
    return torch.optim.AdamW(groups, lr=scaled, betas=(0.9, 0.999), eps=1e-8)


class SupernetTrainer:
    """One process = one GPU.  `step(images, target)` is the body of the reference's hot
    loop; gradient averaging across ranks goes through `reducer` (cream_amd.comm)."""

    def __init__(self, model, optimizer, choices, reducer=None, amp_dtype=torch.bfloat16, max_norm=0.0, mixup_fn=None):
        # mixup_fn: `samples, targets = mixup_fn(samples, targets)` of supernet_engine.py:52-53 (cream_amd.autoformer.data.Mixup:
        # device batches are mixed in place and the soft targets built by ONE launch, csrc/mixup.hip); None: `target` is
        # already the (B, classes) soft target
        self.mixup_fn = mixup_fn
        self.model = model
        self.optimizer = optimizer
        self.choices = choices
        self.reducer = reducer
        self.amp_dtype = amp_dtype
        self.max_norm = max_norm
        self.config = None

    def start_epoch(self, epoch):
        # supernet_engine.py:36 — identical seed on every rank => identical sub-networks
        random.seed(epoch)
        self.model.train()

    def sample(self):
        self.config = sample_configs(self.choices)
        self.model.set_sample_config(self.config)
        return self.config

    def forward_backward(self, images, target):
        dev = images.device.type
        use_amp = self.amp_dtype is not None and self.amp_dtype != torch.float32
        # zero-fill rather than None: under torch 1.7 (the reference's pin) zero_grad()
        # keeps the tensors, so every parameter that was active once keeps receiving
        # weight decay and moment decay
        if self.reducer is not None:
            self.reducer.zero_grad()                 # one memset per bucket
            self.reducer.prepare(self.config)
        else:
            self.optimizer.zero_grad(set_to_none=False)
        with torch.autocast(device_type=dev, dtype=self.amp_dtype, enabled=use_amp):
            logits = self.model(images)
            loss = soft_target_cross_entropy(logits, target)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        return loss

    def step(self, images, target):
        self.sample()                        # (the reference's order: sample, then mixup — supernet_engine.py:43-53)
        if self.mixup_fn is not None:
            images, target = self.mixup_fn(images, target)
        loss = self.forward_backward(images, target)
        if self.max_norm and self.max_norm > 0:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_norm)
        self.optimizer.step()                # (the native optimizer also rewrites the bf16 operand copies)
        return loss


class PendingEval:
    """An evaluation whose kernels are enqueued and whose statistics still live on the device: `result()` is the ONE host
    synchronisation.  A sweep over many sub-networks (evolution search) keeps the device busy across candidates by resolving a
    whole population at once instead of one candidate at a time."""

    def __init__(self, tot, n_total, config, params):
        self._tot, self._n, self._config, self._params, self._done = tot, n_total, config, params, None

    def result(self):
        if self._done is None:
            loss_sum, h1, h5 = self._tot.tolist()
            n = max(float(self._n), 1.0)
            self._done = dict(loss=loss_sum / n, acc1=100.0 * h1 / n, acc5=100.0 * h5 / n, config=self._config, params=self._params)
            self._tot = None
        return self._done


@torch.no_grad()
def evaluate(batches, model, amp_dtype=torch.bfloat16, choices=None, mode='super', retrain_config=None, defer=False):
    """Sub-network evaluation — host-side mirror of `evaluate` in AutoFormer/supernet_engine.py:113-160
    (the inner loop of the evolution search, evolution.py:22-290, and of the validation pass):
    eval mode, ONE sub-network (sampled from `choices` when mode == 'super', else `retrain_config`),
    cross entropy + top-1 / top-5 accuracy averaged over the samples of `batches` (an iterable of
    (images, labels) already on the model's device).  The statistics are accumulated on the device:
    one host synchronisation at the end instead of the reference's three `.item()` per batch.
    Returns {'loss', 'acc1', 'acc5', 'config', 'params'} — or, with defer=True, a `PendingEval` whose `result()` is that dictionary
    (no synchronisation here at all)."""
    model.eval()
    config = sample_configs(choices) if mode == 'super' else retrain_config
    model.set_sample_config(config)
    params = model.get_sampled_params_numel(config)
    dev = next(model.parameters()).device
    tot = torch.zeros(3, dtype=torch.float64, device=dev)        # loss*n, top1 hits, top5 hits (n is host arithmetic: a per-batch
    n_total = 0                                                   # host scalar -> device tensor would be a pageable copy, i.e. a sync)
    use_amp = amp_dtype is not None and amp_dtype != torch.float32 and dev.type == 'cuda'
    for images, labels in batches:
        with torch.autocast(device_type=dev.type, dtype=amp_dtype if use_amp else torch.bfloat16, enabled=use_amp):
            out = model(images)
        out = out.float()
        n = images.shape[0]
        loss = F.cross_entropy(out, labels, reduction='sum')
        top5 = out.topk(min(5, out.shape[1]), dim=1).indices
        hit = top5.eq(labels.view(-1, 1))
        tot += torch.stack([loss.double(), hit[:, 0].sum().double(), hit.any(dim=1).sum().double()])
        n_total += n
    pending = PendingEval(tot, n_total, config, params)
    return pending if defer else pending.result()


# ---- checkpoints: the on-disk format either side of the path ---------------------------
def save_checkpoint(path, model, optimizer=None, lr_scheduler=None, epoch=0, scaler=None, args=None, rank=0):
    """Writes the dictionary of AutoFormer/supernet_train.py:363-370 ('model', 'optimizer',
    'lr_scheduler', 'epoch', 'scaler', 'args'; `utils.save_on_master`: rank 0 only).  The 'model'
    entry has the reference's parameter names and super shapes (the bf16 operand copies of the
    fused path are derived data and are not stored), so the file loads into the reference too."""
    if rank != 0:
        return None
    ckpt = {'model': model.state_dict(), 'epoch': epoch, 'args': args}
    if optimizer is not None:
        ckpt['optimizer'] = optimizer.state_dict()
    if lr_scheduler is not None:
        ckpt['lr_scheduler'] = lr_scheduler.state_dict()
    if scaler is not None:
        ckpt['scaler'] = scaler.state_dict()
    torch.save(ckpt, path)
    return path


def _read_checkpoint(path, trusted):
    """torch.load with tensors-only unpickling; full unpickling (needed for the argparse Namespace in
    the 'args' entry of our own training checkpoints, supernet_train.py:369) only for files the caller
    explicitly trusts — a downloaded supernet-*.pth is never executed as a pickle program."""
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except Exception:
        if not trusted:
            raise
        return torch.load(path, map_location='cpu', weights_only=False)


def load_checkpoint(path_or_dict, model, optimizer=None, lr_scheduler=None, scaler=None, eval_only=False, trusted=False):
    """The resume logic of supernet_train.py:316-330: 'model' always; optimizer / lr_scheduler /
    epoch (+ scaler) only when all three are present and not `eval_only`.  Accepts published
    `supernet-*.pth` files ({'model': state_dict}).  Returns the epoch to START from (0 when the
    file carries no training state).  The bf16 operand copies of the fused blocks are re-derived."""
    ckpt = path_or_dict if isinstance(path_or_dict, dict) else _read_checkpoint(path_or_dict, trusted)
    model.load_state_dict(ckpt['model'])
    _block.refresh_operands(model, force=True)
    start_epoch = 0
    if not eval_only and all(k in ckpt for k in ('optimizer', 'lr_scheduler', 'epoch')):
        if optimizer is not None:
            optimizer.load_state_dict(ckpt['optimizer'])
        if lr_scheduler is not None:
            lr_scheduler.load_state_dict(ckpt['lr_scheduler'])
        start_epoch = ckpt['epoch'] + 1
        if scaler is not None and 'scaler' in ckpt:
            scaler.load_state_dict(ckpt['scaler'])
    return start_epoch
