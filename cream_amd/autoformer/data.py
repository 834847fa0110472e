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


# ---- the uint8 -> normalised float transform of a batch on the device (lib/datasets.py:189-220) ---------------------------------
IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)       # timm.data.constants, imported at lib/datasets.py:10
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def eval_crop_params(height, width, input_size=224):
    """`Resize(int(256 / 224 * input_size), interpolation=3)` + `CenterCrop(input_size)` (lib/datasets.py:211-216) as
    (box, resized, window): torchvision's F.resize sends the shorter side to `size` and the other to int(size * long / short);
    F.center_crop starts at int(round((h - th) / 2.))."""
    size = int((256 / 224) * input_size)
    if width <= height:
        rw, rh = size, int(size * height / width)
    else:
        rh, rw = size, int(size * width / height)
    return (0, 0, height, width), (rh, rw), (int(round((rh - input_size) / 2.)), int(round((rw - input_size) / 2.)))


def train_crop_params(height, width, rng, input_size=224, scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.), hflip=0.5):
    """The geometric head of timm's training transform (create_transform(is_training=True), lib/datasets.py:193-202):
    RandomResizedCropAndInterpolation.get_params followed by RandomHorizontalFlip, drawn from `rng` (a `random.Random`) in
    that order of calls.  timm is third-party and not vendored in the reference: restated from its published definition
    (ten attempts of area x aspect sampling, then the central crop clamped to the ratio range) — parity unpinned by the
    reference; the RESAMPLING of the crop is pinned against Pillow.  -> (box, resized, window, flip)."""
    area = width * height
    box = None
    for _ in range(10):
        target_area = rng.uniform(*scale) * area
        aspect = math.exp(rng.uniform(math.log(ratio[0]), math.log(ratio[1])))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if w <= width and h <= height:
            top = rng.randint(0, height - h)
            left = rng.randint(0, width - w)
            box = (top, left, h, w)
            break
    if box is None:
        in_ratio = width / height
        if in_ratio < min(ratio):
            w = width
            h = int(round(w / min(ratio)))
        elif in_ratio > max(ratio):
            h = height
            w = int(round(h * max(ratio)))
        else:
            w, h = width, height
        box = ((height - h) // 2, (width - w) // 2, h, w)
    flip = rng.random() < hflip
    return box, (input_size, input_size), (0, 0), flip


def random_erasing_params(rng, height=224, width=224, probability=0.25, min_area=0.02, max_area=1 / 3, min_aspect=0.3, count=1):
    """The box of timm's RandomErasing (create_transform(re_prob=args.reprob, re_mode='pixel', re_count=1), lib/datasets.py:199-201)
    for ONE image, drawn from `rng` (a `random.Random`) in timm's order of calls: the probability gate, then up to ten attempts of
    area x aspect sampling.  Third-party, restated from its published definition (parity unpinned by the reference).
    -> (top, left, h, w, seed) or None; `seed` names the noise the device writes into the box."""
    if rng.random() > probability:
        return None
    area = height * width
    log_lo, log_hi = math.log(min_aspect), math.log(1 / min_aspect)
    for _ in range(10):
        target_area = rng.uniform(min_area, max_area) * area / count
        aspect = math.exp(rng.uniform(log_lo, log_hi))
        h = int(round(math.sqrt(target_area * aspect)))
        w = int(round(math.sqrt(target_area / aspect)))
        if w < width and h < height:
            top = rng.randint(0, height - h)
            left = rng.randint(0, width - w)
            return top, left, h, w, rng.getrandbits(32)
    return None


def erase_noise_reference(seed, C, H, W):
    """numpy restatement of the device's noise (csrc/image_transform.hip: erase_noise) as a (C, H, W) float32 array — for the tests."""
    import numpy as np

    def mix32(x):
        x = x.astype(np.uint64)
        x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & np.uint64(0xFFFFFFFF)
        x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & np.uint64(0xFFFFFFFF)
        x ^= x >> np.uint64(16)
        return x

    c = np.arange(C, dtype=np.uint64)[:, None, None]
    y = np.arange(H, dtype=np.uint64)[None, :, None]
    x = np.arange(W, dtype=np.uint64)[None, None, :]
    key = mix32(np.uint64(seed & 0xFFFFFFFF) ^ (((c + np.uint64(1)) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)))
    h1 = mix32(key ^ ((y << np.uint64(16)) | x))
    h2 = mix32(h1 ^ np.uint64(0x85EBCA6B))
    u1 = ((h1 >> np.uint64(8)).astype(np.float32) + np.float32(1)) * np.float32(1 / 16777216)
    u2 = (h2 >> np.uint64(8)).astype(np.float32) * np.float32(1 / 16777216)
    return (np.sqrt(np.float32(-2) * np.log(u1)) * np.cos(np.float32(6.28318530717958647692) * u2)).astype(np.float32)


class DeviceTransform:
    """A batch of decoded frames (HWC uint8 RGB arrays of any sizes) -> (B, 3, S, S) float32 on the device, as the reference's
    per-image transforms produce it: F.crop -> Pillow's bicubic F.resize -> window (CenterCrop) -> mirror -> ToTensor -> Normalize,
    three launches for the whole batch (cream_image_batch_transform; byte-exact with Pillow's resize, bit-exact float ops).
    The frames travel as ONE packed uint8 buffer (each frame at a 4-byte aligned offset) from pinned memory; RandAugment /
    RandomErasing of the training recipe stay host-side (out of scope)."""

    def __init__(self, input_size=224, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD, device="cuda"):
        import ctypes
        self.size, self.device = int(input_size), torch.device(device)
        self._mean = (ctypes.c_float * 3)(*mean)
        self._std = (ctypes.c_float * 3)(*std)
        self._ws = None
        self._pool = None                      # packing threads (numpy's large copies release the GIL)
        self.pack_threads = 8

    def plan(self, shapes, params):
        """shapes: [(H, W)], params: [(box, resized, window[, flip[, erase]])] with erase = None | (top, left, h, w, seed)
        (random_erasing_params) -> (ImageDesc array (planned), packed byte count, workspace bytes)."""
        from .. import _lib
        B = len(shapes)
        descs = (_lib.ImageDesc * B)()
        off = 0
        for d, (h, w), p in zip(descs, shapes, params):
            box, resized, window = p[0], p[1], p[2]
            d.offset, d.height, d.width, d.row_stride = off, h, w, 3 * w
            d.box_top, d.box_left, d.box_h, d.box_w = box
            d.resized_h, d.resized_w = resized
            d.win_top, d.win_left = window
            d.flip = 1 if (len(p) > 3 and p[3]) else 0
            if len(p) > 4 and p[4] is not None:
                d.erase_top, d.erase_left, d.erase_h, d.erase_w, d.erase_seed = p[4]
            off += (h * w * 3 + 3) // 4 * 4
        ws = _lib.load().cream_image_batch_plan(descs, B, self.size, self.size)
        if ws < 0:
            _lib.check(int(ws), "cream_image_batch_plan")
        return descs, off, int(ws)

    def __call__(self, frames, params):
        """frames: list of (H, W, 3) uint8 numpy arrays / tensors; params as for `plan`."""
        import ctypes
        import numpy as np
        from .. import _lib
        B = len(frames)
        shapes = [tuple(f.shape[:2]) for f in frames]
        descs, nbytes, ws_bytes = self.plan(shapes, params)
        packed = torch.empty(nbytes, dtype=torch.uint8, pin_memory=self.device.type == "cuda")
        pk = packed.numpy()

        def put(i):
            f = frames[i]
            a = np.ascontiguousarray(f.numpy() if isinstance(f, torch.Tensor) else f, dtype=np.uint8)
            pk[descs[i].offset:descs[i].offset + a.size] = a.reshape(-1)

        if B >= 16 and self.pack_threads > 1:   # one copy per frame into the pinned buffer, spread over a few threads
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=self.pack_threads)
            list(self._pool.map(put, range(B)))
        else:
            for i in range(B):
                put(i)
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
        if self.device.type == "cuda":
            raw = raw.pin_memory()
        with torch.cuda.device(self.device):
            pix = packed.to(self.device, non_blocking=True)
            dd = raw.to(self.device, non_blocking=True)
            if self._ws is None or self._ws.numel() < ws_bytes:
                self._ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=self.device)
            out = torch.empty((B, 3, self.size, self.size), dtype=torch.float32, device=self.device)
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(_lib.load().cream_image_batch_transform(
                ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(pix.data_ptr()), nbytes, descs, ctypes.c_void_p(dd.data_ptr()), B,
                self.size, self.size, self._mean, self._std, ctypes.c_void_p(self._ws.data_ptr()), self._ws.numel(), st),
                "cream_image_batch_transform")
            # the staging tensors are read by the copies / kernels enqueued above
            pix.record_stream(torch.cuda.current_stream())
            dd.record_stream(torch.cuda.current_stream())
        return out


class DeviceBatches:
    """`batches` for engine.evaluate / the train step from a loader of DECODED frames: wraps an iterable of
    (frames: list of (H, W, 3) uint8 arrays, labels: int sequence) and yields (samples (B, 3, S, S) fp32 on the device, labels int64
    on the device) — what `DataLoader(ImageFolder(root, transform=build_transform(...)))` yields in the reference (lib/datasets.py:
    222-239, supernet_train.py:222-241), with the transform moved behind the loader and onto the device.
    mode 'eval': Resize + CenterCrop; mode 'train': RandomResizedCrop + flip (+ RandomErasing with `reprob` > 0) from `rng`."""

    def __init__(self, loader, transform, mode="eval", rng=None, reprob=0.0):
        import random as _random
        if mode not in ("eval", "train"):
            raise ValueError(f"unknown mode {mode!r}")
        self.loader, self.transform, self.mode, self.reprob = loader, transform, mode, float(reprob)
        self.rng = rng if rng is not None else _random.Random(0)

    def params_for(self, shapes):
        size = self.transform.size
        if self.mode == "eval":
            return [eval_crop_params(h, w, size) + (False, None) for h, w in shapes]
        out = []
        for h, w in shapes:
            box, resized, window, flip = train_crop_params(h, w, self.rng, size)
            erase = random_erasing_params(self.rng, size, size, self.reprob) if self.reprob > 0 else None
            out.append((box, resized, window, flip, erase))
        return out

    def __iter__(self):
        for frames, labels in self.loader:
            shapes = [tuple(f.shape[:2]) for f in frames]
            samples = self.transform(frames, self.params_for(shapes))
            yield samples, torch.as_tensor(labels, dtype=torch.int64).to(samples.device, non_blocking=True)


# ---- the dataset / loader either side of the device transform (lib/datasets.py:152-187, supernet_train.py:222-243) ----------------
IMG_EXTENSIONS = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')    # torchvision.datasets.folder


class ImageFolderFrames(torch.utils.data.Dataset):
    """`datasets.ImageFolder(root)` of the reference's `build_dataset` (data_set 'IMNET' / 'EVO_IMNET': lib/datasets.py:170-177)
    with the transform moved behind the loader: class = sub-directory (sorted names -> indices), samples = the image files below
    it in sorted walk order (torchvision's `make_dataset`), `__getitem__` = the DECODED frame (`default_loader`: PIL open +
    convert('RGB')) as an (H, W, 3) uint8 array and the class index.  Decoding is host work for the loader's worker processes
    (out of scope for the device); everything after it is `DeviceTransform`."""

    def __init__(self, root):
        import os
        self.root = root
        self.classes = sorted(e.name for e in os.scandir(root) if e.is_dir())
        if not self.classes:
            raise FileNotFoundError(f"Couldn't find any class folder in {root}.")
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []
        for c in self.classes:
            for base, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(base, f), self.class_to_idx[c]))
        self.targets = [t for _, t in self.samples]

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        import numpy as np
        from PIL import Image
        path, target = self.samples[i]
        with open(path, 'rb') as fh:
            frame = np.asarray(Image.open(fh).convert('RGB'))
        return frame, target


def collate_frames(batch):
    """Frames have different sizes: a batch is (list of frames, list of labels) — what `DeviceBatches` takes."""
    return [b[0] for b in batch], [b[1] for b in batch]


def frame_loader(dataset, batch_size, sampler=None, num_workers=0, drop_last=False):
    """The `DataLoader` of supernet_train.py:231-243 over a dataset of decoded frames (pin_memory is `DeviceTransform`'s job:
    it packs the frames into ONE pinned buffer per batch)."""
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers,
                                       drop_last=drop_last, collate_fn=collate_frames)
