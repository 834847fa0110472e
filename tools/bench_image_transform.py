"""The device input transform (cream_image_batch_transform) on a batch of 128 ImageNet-shaped frames: kernel time of the two launches
(HIP events on the launch stream, frames already resident in HBM), the end-to-end call with the host packing and the PCIe copy, and
Pillow + torch on the host cores beside it (the reference's per-image path, one thread).
    python tools/bench_image_transform.py [eval|train]   ->  one JSON line"""
import ctypes
import json
import random
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cream_amd import _lib                                   # noqa: E402
from cream_amd.autoformer import data as D                   # noqa: E402


def main():
    pipeline = sys.argv[1] if len(sys.argv) > 1 else "train"
    B, size, dev = 128, 224, "cuda:0"
    rng = np.random.default_rng(0)
    pr = random.Random(0)
    # ImageNet's typical frames: 500 x 375 / 375 x 500 / 500 x 333, a few larger ones
    mix = [(375, 500), (500, 375), (333, 500), (480, 640), (768, 1024)]
    if len(sys.argv) > 2 and sys.argv[2] == "typical":            # ImageNet's three most common frame sizes only
        mix = mix[:3]
    shapes = [mix[i % len(mix)] for i in range(B)]
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    params = [D.eval_crop_params(h, w) + (False,) if pipeline == "eval" else D.train_crop_params(h, w, pr) for h, w in shapes]
    T = D.DeviceTransform(size, device=dev)
    out = T(frames, params)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = T(frames, params)
    torch.cuda.synchronize()
    e2e = (time.perf_counter() - t0) / 5
    # kernels only: frames and descriptors resident
    descs, nbytes, ws = T.plan(shapes, params)
    pix = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    for d, f in zip(descs, frames):
        pix[d.offset:d.offset + f.size] = torch.from_numpy(f.reshape(-1)).to(dev)
    dd = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
    wsb = torch.empty(ws, dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    lib = _lib.load()
    st = torch.cuda.current_stream()
    call = lambda: lib.cream_image_batch_transform(p(out), p(pix), nbytes, descs, p(dd), B, size, size, T._mean, T._std, p(wsb), ws,
                                                   ctypes.c_void_p(st.cuda_stream))
    for _ in range(3):
        assert call() == 0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record(st)
    for _ in range(20):
        call()
    ev[1].record(st)
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    box_bytes = sum(d.nrows * d.box_w * 3 for d in descs)
    algo = box_bytes + 2 * sum(d.nrows * size * 3 for d in descs) + B * 3 * size * size * 4
    # the host path on one core: Pillow's crop + resize + crop, then the two float ops
    from PIL import Image
    t0 = time.perf_counter()
    n = 32
    for f, (box, resized, window, flip) in list(zip(frames, params))[:n]:
        t, l, h, w = box
        im = Image.fromarray(f).crop((l, t, l + w, t + h)).resize((resized[1], resized[0]), Image.BICUBIC)
        im = im.crop((window[1], window[0], window[1] + size, window[0] + size))
        if flip:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        x = torch.from_numpy(np.array(im)).permute(2, 0, 1).float().div(255)
        x = x.sub(torch.tensor(D.IMAGENET_DEFAULT_MEAN).view(3, 1, 1)).div(torch.tensor(D.IMAGENET_DEFAULT_STD).view(3, 1, 1))
    cpu = (time.perf_counter() - t0) / n
    print(json.dumps({"workload": f"input transform ({pipeline}): {B} decoded frames ({min(s[0] for s in shapes)}x{min(s[1] for s in shapes)} .. {max(s[0] for s in shapes)}x{max(s[1] for s in shapes)}) -> (128, 3, 224, 224) fp32",
                      "kernels_ms": round(ms, 4), "images_per_sec_kernels": round(B / ms * 1e3), "algorithmic_bytes": algo,
                      "GBps": round(algo / ms / 1e6, 1), "frac_of_8TBps": round(algo / ms / 1e6 / 8000, 3),
                      "end_to_end_ms_with_host_packing_and_pcie": round(e2e * 1e3, 3),
                      "images_per_sec_end_to_end": round(B / e2e), "packed_MB": round(nbytes / 1e6, 1),
                      "cpu_pillow_torch_ms_per_image_1_core": round(cpu * 1e3, 3), "cpu_images_per_sec_1_core": round(1 / cpu, 1)}))


if __name__ == "__main__":
    main()
