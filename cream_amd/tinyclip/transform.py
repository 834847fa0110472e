"""TinyCLIP's image preprocessing on the device — mirror of `image_transform` in TinyCLIP/src/open_clip/transform.py:71-122.

    train:  RandomResizedCrop(image_size, scale=(0.9, 1.0), interpolation=BICUBIC) -> ToTensor -> Normalize(OPENAI mean / std)
    val  :  Resize(image_size, BICUBIC) -> CenterCrop(image_size) -> ToTensor -> Normalize     (val_keep_ratio=True, the default)
            Resize((image_size, image_size), BICUBIC) -> ToTensor -> Normalize                  (val_keep_ratio=False)

Unlike AutoFormer's recipe there is no PIL-level augmentation between the crop and `ToTensor`: BOTH pipelines run on the device end
to end behind the decoder (`cream_amd.autoformer.data.DeviceTransform` with CLIP's statistics: crop -> Pillow's bicubic resize,
byte-exact -> window -> ToTensor -> Normalize, three launches per batch).  The crop parameters are torchvision's
`RandomResizedCrop.get_params` (third-party, not vendored in the reference: restated from its published definition, drawn from
torch's generator in its order of calls — `torch.manual_seed` replays them; parity unpinned by the reference).
`resize_longest_max` (letterboxing with a fill colour) is not covered."""
import math

import torch

from ..autoformer.data import DeviceTransform

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)        # TinyCLIP/src/open_clip/constants.py:1-2
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


def train_crop_params(height, width, image_size=224, scale=(0.9, 1.0), ratio=(3. / 4., 4. / 3.), generator=None):
    """transform.py:96-97 -> (box, resized, window, flip=False)."""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1], generator=generator).item()
        aspect = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1], generator=generator)).item()
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,), generator=generator).item()
            j = torch.randint(0, width - w + 1, size=(1,), generator=generator).item()
            return (i, j, h, w), (image_size, image_size), (0, 0), False
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return ((height - h) // 2, (width - w) // 2, h, w), (image_size, image_size), (0, 0), False


def val_crop_params(height, width, image_size=224, keep_ratio=True):
    """transform.py:108-117: the shorter side to image_size (the other int(image_size * long / short)), centre crop — or the whole
    frame squeezed to image_size x image_size."""
    if not keep_ratio:
        return (0, 0, height, width), (image_size, image_size), (0, 0), False
    if width <= height:
        rw, rh = image_size, int(image_size * height / width)
    else:
        rh, rw = image_size, int(image_size * width / height)
    return ((0, 0, height, width), (rh, rw), (int(round((rh - image_size) / 2.)), int(round((rw - image_size) / 2.))), False)


def device_transform(image_size=224, mean=None, std=None, device="cuda"):
    """`image_transform(image_size, is_train, mean, std)` as a batch operation on the device: call with (frames, params)."""
    return DeviceTransform(image_size, mean or OPENAI_DATASET_MEAN, std or OPENAI_DATASET_STD, device=device)
