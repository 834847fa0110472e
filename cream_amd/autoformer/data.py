"""Host side of the input path next to the train step — mirror of AutoFormer/lib/samplers.py:6-57 (`RASampler`, the
repeated-augmentation sampler the reference trains with: `--repeated-aug` is its default, supernet_train.py:118-122,
209-213).

Same constructor, `set_epoch`, `__len__` and, for every (dataset length, replicas, rank, epoch, shuffle), the SAME index
sequence: each sample appears three times in a row in the epoch's permutation, the list is padded with its own head to a
multiple of the replica count, rank r takes every `num_replicas`-th entry from r (so the three copies of a sample go to
three different ranks, where they get three different augmentations), and an epoch is cut to
floor(len // 256 * 256 / replicas) draws.  The reference materialises the 3 x len Python list on every rank and epoch
(3.8 M boxed ints for ImageNet); here the rank's entries are computed directly as a tensor expression of the permutation.
"""
import math

import torch
import torch.distributed as dist


class RASampler(torch.utils.data.Sampler):
    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        if num_replicas is None:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("Requires distributed package to be available")
            num_replicas = dist.get_world_size()
        if rank is None:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("Requires distributed package to be available")
            rank = dist.get_rank()
        self.dataset, self.num_replicas, self.rank, self.epoch, self.shuffle = dataset, num_replicas, rank, 0, shuffle
        n = len(dataset)
        self.num_samples = int(math.ceil(n * 3.0 / num_replicas))
        self.total_size = self.num_samples * num_replicas
        self.num_selected_samples = int(math.floor(n // 256 * 256 / num_replicas))

    def indices(self):
        """The rank's draws of this epoch as an int64 tensor."""
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            perm = torch.randperm(n, generator=g)
        else:
            perm = torch.arange(n)
        # position t of the repeated, padded list holds perm[(t mod 3n) // 3]; this rank reads t = rank, rank + R, ...
        t = torch.arange(self.rank, self.total_size, self.num_replicas)[:self.num_selected_samples]
        return perm[(t % (3 * n)) // 3] if n > 0 else perm

    def __iter__(self):
        return iter(self.indices().tolist())

    def __len__(self):
        return self.num_selected_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class Mixup:
    """Batch-mode Mixup / CutMix with label smoothing as the step's `mixup_fn(samples, targets)` (supernet_engine.py:52-53;
    constructed at supernet_train.py:245-251: mixup 0.8, cutmix 1.0, prob 1.0, switch_prob 0.5, mode 'batch', smoothing 0.1).
    timm.data.Mixup is third-party code that the reference does not vendor: restated here from its published definition —
    parity UNPINNED (no reference-made fixture can exist) — with numpy's global RNG in timm's draw order (apply?, cutmix?,
    lambda ~ Beta, box centre y, x), so that a seeded run makes the same decisions.  Works in place on device tensors:
        mixup : x <- lam x + (1 - lam) x.flip(0)
        cutmix: x[:, :, box] <- x.flip(0)[:, :, box],  lam <- 1 - box area / image area
        target: lam onehot_s(y) + (1 - lam) onehot_s(y.flip(0)),  onehot_s = smoothing / C off, 1 - smoothing + off on."""

    def __init__(self, mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, mode='batch', label_smoothing=0.1,
                 num_classes=1000):
        assert mode == 'batch', "the recipe's mode (supernet_train.py:249); 'pair' / 'elem' are not on this path"
        self.mixup_alpha, self.cutmix_alpha = mixup_alpha, cutmix_alpha
        self.mix_prob, self.switch_prob = prob, switch_prob
        self.label_smoothing, self.num_classes = label_smoothing, num_classes
        self.mixup_enabled = True

    def _params_per_batch(self):
        import numpy as np
        lam, use_cutmix = 1.0, False
        if self.mixup_enabled and np.random.rand() < self.mix_prob:
            if self.mixup_alpha > 0. and self.cutmix_alpha > 0.:
                use_cutmix = np.random.rand() < self.switch_prob
                lam = np.random.beta(self.cutmix_alpha, self.cutmix_alpha) if use_cutmix else \
                    np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.mixup_alpha > 0.:
                lam = np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.cutmix_alpha > 0.:
                use_cutmix = True
                lam = np.random.beta(self.cutmix_alpha, self.cutmix_alpha)
        return float(lam), use_cutmix

    @staticmethod
    def rand_bbox(height, width, lam):
        import numpy as np
        ratio = (1.0 - lam) ** 0.5
        cut_h, cut_w = int(height * ratio), int(width * ratio)
        cy, cx = np.random.randint(0, height), np.random.randint(0, width)
        yl, yh = int(np.clip(cy - cut_h // 2, 0, height)), int(np.clip(cy + cut_h // 2, 0, height))
        xl, xh = int(np.clip(cx - cut_w // 2, 0, width)), int(np.clip(cx + cut_w // 2, 0, width))
        return yl, yh, xl, xh

    def mix_targets(self, target, lam):
        off = self.label_smoothing / self.num_classes
        on = 1.0 - self.label_smoothing + off
        y = torch.full((target.shape[0], self.num_classes), off, device=target.device, dtype=torch.float32)
        y.scatter_(1, target.long().view(-1, 1), on)
        return y * lam + y.flip(0) * (1.0 - lam)

    def _native(self, x, target, lam, use_cutmix):
        """Device tensors: images mixed in place and the soft targets built by ONE launch (csrc/mixup.hip)."""
        import ctypes
        from .. import _lib
        B, C, H, W = x.shape
        box = (0, 0, 0, 0)
        if lam != 1.0 and use_cutmix:
            box = self.rand_bbox(H, W, lam)
            lam = 1.0 - (box[1] - box[0]) * (box[3] - box[2]) / float(H * W)                  # corrected for the clipped box
        y = torch.empty((B, self.num_classes), device=x.device, dtype=torch.float32)
        t = target.to(torch.int64).contiguous()
        with torch.cuda.device(x.device):
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(_lib.load().cream_mixup_cutmix(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                                      ctypes.c_void_p(t.data_ptr()), B, C, H, W, self.num_classes, float(lam),
                                                      1 if use_cutmix else 0, box[0], box[1], box[2], box[3],
                                                      float(self.label_smoothing), st), "cream_mixup_cutmix")
        return x, y

    def __call__(self, x, target):
        assert x.shape[0] % 2 == 0, 'Batch size should be even when using this'
        lam, use_cutmix = self._params_per_batch()
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and x.shape[-1] % 4 == 0
                and x.data_ptr() % 16 == 0 and target.dim() == 1):
            return self._native(x, target, lam, use_cutmix)
        if lam != 1.0:
            if use_cutmix:
                yl, yh, xl, xh = self.rand_bbox(x.shape[-2], x.shape[-1], lam)
                lam = 1.0 - (yh - yl) * (xh - xl) / float(x.shape[-2] * x.shape[-1])     # corrected for the clipped box
                x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh]
            else:
                x.copy_(x * lam + x.flip(0) * (1.0 - lam))
        return x, self.mix_targets(target, lam)
