"""One affinity-mimicking distillation step of TinyCLIP — the body of `train_one_epoch`
(TinyCLIP/src/training/train.py:84-560) for the dense student / frozen teacher recipe of the shipped scripts
(`--distillation --local-loss --gather-with-grad --logit-scale 50 --norm_gradient_clip 5`, script/*.sh):

    teacher (no grad, autocast): normalised image / text features, logit scale fixed to `logit_scale`     (:405-420)
    student (autocast):          normalised features, exp(logit scale)                                   (:227-242)
    loss = 0.5 w (soft CE image->text) + 0.5 w (soft CE text->image),  w = alpha * distillation weight   (:173-186)
           [+ (1 - alpha) * weight * the hard CLIP loss when alpha < 1]                                   (:192-195)
    backward; clip the global gradient norm; optimizer step                                              (:505-513)
    student logit scale: filled with ln(logit_scale) when that flag is given, else clamped to [0, ln 100] (:526-530)

bf16 autocast needs no loss scaler (the reference's `amp` = fp16 + GradScaler; `amp_bfloat16` is its own bf16 mode,
precision.py:7-11).  Multi-GPU: one process per GPU; the only exchanges are the feature gathers inside `ClipSoftLoss`
(cream_amd/tinyclip/soft_loss.py) and the gradient average of `GradReducer`.
"""
import math

import torch
import torch.nn.functional as F

from .soft_loss import ClipSoftLoss, gather_features_with_grad


def hard_clip_loss(image_features, text_features, logit_scale, rank=0, group=None):
    """open_clip/loss.py `ClipLoss` with local_loss + gather_with_grad: contrastive CE against the global batch."""
    all_image, all_text = gather_features_with_grad(image_features, text_features, True, group)
    li = logit_scale * image_features @ all_text.T
    lt = logit_scale * text_features @ all_image.T
    n = image_features.shape[0]
    labels = torch.arange(n, device=li.device) + n * rank
    return (F.cross_entropy(li, labels) + F.cross_entropy(lt, labels)) / 2


def tower_bucket_of(name, blocks_per_bucket=4):
    """Bucket key of a student parameter: the reference wraps the image tower, the text tower and the logit scale in three
    separate DDP instances (`ddpify`, open_clip/model.py:977-988), so each tower's gradients travel as soon as THAT tower's
    backward has produced them.  Here: one GradReducer whose buckets never cross a tower — `<tower>.tr<k>` for every
    `blocks_per_bucket` consecutive transformer blocks (a ViT-39M/16 block is ~7 MB of fp32 gradients: 4 per message keeps
    the xGMI links busy without waiting for the whole tower), `<tower>.rest` for embeddings / projections / final norm."""
    parts = name.split(".")
    tower = parts[0]
    if "resblocks" in parts:
        return f"{tower}.tr{int(parts[parts.index('resblocks') + 1]) // blocks_per_bucket:02d}"
    return tower if tower == "_logit_scale" else f"{tower}.rest"


def make_reducer(student, group=None, mode="allreduce", blocks_per_bucket=4):
    """`ddpify` of the step (model.py:977-988): per-tower gradient buckets on the side-stream reducer (cream_amd/comm.py).  No
    active-slice packing: TinyCLIP's towers are dense."""
    from ..comm import GradReducer
    return GradReducer(student, process_group=group, mode=mode, slice_of=None,
                       bucket_of=lambda n: tower_bucket_of(n, blocks_per_bucket))


class DistillStep:
    def __init__(self, student, teacher, optimizer, *, logit_scale=50.0, distillation_alpha=1.0, distillation_weight=1.0,
                 norm_gradient_clip=5.0, amp_dtype=torch.bfloat16, rank=0, world_size=1, group=None, reducer=None):
        self.student, self.teacher, self.optimizer = student, teacher, optimizer
        self.logit_scale, self.alpha, self.weight, self.clip = logit_scale, distillation_alpha, distillation_weight, norm_gradient_clip
        self.amp_dtype, self.rank, self.group, self.reducer = amp_dtype, rank, group, reducer
        self.soft_loss = ClipSoftLoss(local_loss=True, gather_with_grad=True, rank=rank, world_size=world_size, group=group)
        for p in teacher.parameters():
            p.requires_grad = False
        teacher.eval()
        self.params = [p for p in student.parameters() if p.requires_grad]

    def _autocast(self, device):
        on = self.amp_dtype != torch.float32 and device.type == "cuda"
        return torch.autocast("cuda", dtype=self.amp_dtype if on else torch.bfloat16, enabled=on)

    def teacher_outputs(self, images, texts):
        with torch.no_grad(), self._autocast(images.device):
            if self.logit_scale is not None:
                self.teacher.logit_scale.fill_(math.log(self.logit_scale))                 # train.py:408-409
            ti = F.normalize(self.teacher.encode_image(images), dim=-1)
            tt = F.normalize(self.teacher.encode_text(texts), dim=-1)
            return ti, tt, self.teacher.logit_scale.exp()

    def loss(self, images, texts):
        ti, tt, ts = self.teacher_outputs(images, texts)
        with self._autocast(images.device):
            fi, ft, s = self.student(images, texts, normalized=True)
            total = 0.0
            if self.alpha > 0.0 and self.weight > 0.0:
                i2t, t2i = self.soft_loss(fi, ft, s, ti, tt, ts, average_two_losses=False)
                total = total + 0.5 * self.alpha * self.weight * (i2t + t2i)
            if self.alpha < 1.0 and self.weight > 0.0:
                total = total + (1.0 - self.alpha) * self.weight * hard_clip_loss(fi, ft, s, self.rank, self.group)
        return total

    def step(self, images, texts):
        if self.reducer is not None:
            # the reducer owns the gradients (views into its flat arena): zero-fill them in place and arm the buckets —
            # zero_grad(set_to_none=True) would detach every p.grad from the arena and no collective would ever run
            self.reducer.zero_grad()
            self.reducer.prepare(None)
        else:
            # gradients are zero-filled in place (one multi-tensor launch) rather than dropped: the native tower nodes ADD
            # their weight gradients into existing .grad tensors, and re-creating ~400 of them per step costs ~400 fills
            self.optimizer.zero_grad(set_to_none=False)
        loss = self.loss(images, texts)
        loss.backward()
        if self.reducer is not None:
            assert self.reducer.owns_grads(), "a parameter gradient no longer aliases the reducer's arena"
            self.reducer.finish()
        if self.clip is not None:
            torch.nn.utils.clip_grad_norm_(self.params, self.clip, norm_type=2.0)
        self.optimizer.step()
        with torch.no_grad():                                                              # train.py:526-530
            if self.logit_scale is not None:
                self.student.logit_scale.fill_(math.log(self.logit_scale))
            else:
                self.student.logit_scale.clamp_(0, math.log(100))
        return loss.detach()
