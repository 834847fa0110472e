"""Golden vectors of the input transform, made with the library the reference's transforms call: Pillow.
    python tests/golden/make_image_transform_golden.py        ->  tests/golden/image_transform.npz
For every case: a seeded random uint8 frame, the (box, resized, window, flip) of one of the two pipelines of
AutoFormer/lib/datasets.py:189-220, and what torchvision's functional ops produce on a PIL image —
    F.crop = Image.crop, F.resize = Image.resize(size, BICUBIC), F.center_crop = Image.crop, F.hflip = transpose(FLIP_LEFT_RIGHT)
— as uint8 (before ToTensor / Normalize, which are exact float32 ops checked against torch in the tests).
Pillow version used is recorded in the file."""
import os
import random
import sys

import numpy as np
import PIL
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from cream_amd.autoformer.data import eval_crop_params, train_crop_params  # noqa: E402

# (height, width, pipeline, input_size)
CASES = [(48, 64, "eval", 32), (64, 48, "eval", 32), (37, 53, "train", 32), (120, 90, "train", 32), (33, 33, "eval", 32),
         (200, 301, "train", 64), (75, 100, "eval", 64), (30, 40, "train", 64)]


def pil_pipeline(frame, box, resized, window, out, flip):
    t, l, h, w = box
    im = Image.fromarray(frame).crop((l, t, l + w, t + h)).resize((resized[1], resized[0]), Image.BICUBIC)
    im = im.crop((window[1], window[0], window[1] + out, window[0] + out))
    if flip:
        im = im.transpose(Image.FLIP_LEFT_RIGHT)
    return np.asarray(im)


def main():
    rng = np.random.default_rng(20260930)
    pr = random.Random(7)
    out = {"pillow_version": np.array(PIL.__version__)}
    for i, (h, w, pipe, size) in enumerate(CASES):
        frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if pipe == "eval":
            box, resized, window = eval_crop_params(h, w, size)
            flip = False
        else:
            box, resized, window, flip = train_crop_params(h, w, pr, size)
        out[f"frame{i}"] = frame
        out[f"params{i}"] = np.array(list(box) + list(resized) + list(window) + [int(flip), size], dtype=np.int64)
        out[f"u8_{i}"] = pil_pipeline(frame, box, resized, window, size, flip)
    np.savez_compressed(os.path.join(HERE, "image_transform.npz"), **out)
    print("wrote", len(CASES), "cases; Pillow", PIL.__version__)


if __name__ == "__main__":
    main()
