"""GPU: the device input transform (cream_image_batch_transform, called through the C ABI by autoformer.data.DeviceTransform) against
the oracle's restatement of Pillow's resize + torchvision's ToTensor / Normalize, against the Pillow-made fixtures and against Pillow
itself: BIT-EXACT (integer resampling; two IEEE float32 divisions)."""
import random

import numpy as np
import pytest
import torch

from oracle import image_transform_oracle as O
from cream_amd.autoformer import data as D

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = np.load(__file__.rsplit("/", 1)[0] + "/golden/image_transform.npz")
NCASES = sum(1 for k in GOLD.files if k.startswith("frame"))


def _ref(frame, box, resized, window, size, flip):
    return O.to_tensor_normalize(O.resized_window(frame, box, resized, window, (size, size), flip))


def test_fixtures_bit_exact_as_one_ragged_batch_per_size():
    by_size = {}
    for i in range(NCASES):
        p = [int(v) for v in GOLD[f"params{i}"]]
        by_size.setdefault(p[9], []).append((GOLD[f"frame{i}"], (tuple(p[0:4]), tuple(p[4:6]), tuple(p[6:8]), bool(p[8])), GOLD[f"u8_{i}"]))
    for size, cases in by_size.items():
        T = D.DeviceTransform(size, device=DEV)
        out = T([c[0] for c in cases], [c[1] for c in cases]).cpu()
        for o, (frame, prm, u8) in zip(out, cases):
            assert torch.equal(o, O.to_tensor_normalize(u8))            # the Pillow-made bytes, then the two float ops
            assert torch.equal(o, _ref(frame, *prm[:3], size, prm[3]))


@pytest.mark.parametrize("pipeline", ["eval", "train"])
def test_imagenet_shaped_batch_bit_exact_against_the_oracle_and_pillow(pipeline):
    """Ragged frames of ImageNet-like sizes (up- and down-scaling, portrait / landscape, a tiny and a large one) -> 224 x 224."""
    rng = np.random.default_rng(11)
    pr = random.Random(5)
    shapes = [(375, 500), (500, 375), (333, 500), (224, 224), (64, 80), (768, 1024), (1200, 900), (100, 400)]
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    if pipeline == "eval":
        params = [D.eval_crop_params(h, w) + (False,) for h, w in shapes]
    else:
        params = [D.train_crop_params(h, w, pr) for h, w in shapes]
    out = D.DeviceTransform(224, device=DEV)(frames, params).cpu()
    try:
        from PIL import Image
    except ImportError:
        Image = None
    for o, f, (box, resized, window, flip) in zip(out, frames, params):
        assert torch.equal(o, _ref(f, box, resized, window, 224, flip))
        if Image is not None:
            t, l, h, w = box
            im = Image.fromarray(f).crop((l, t, l + w, t + h)).resize((resized[1], resized[0]), Image.BICUBIC)
            im = im.crop((window[1], window[0], window[1] + 224, window[0] + 224))
            if flip:
                im = im.transpose(Image.FLIP_LEFT_RIGHT)
            assert torch.equal(o, O.to_tensor_normalize(np.asarray(im)))


def test_properties_at_batch_128():
    """A constant frame stays constant through both passes (the fixed-point coefficients of a row sum to 2^22 within the rounding
    Pillow itself has: checked against the oracle, not assumed); the mirror flag is a pure column reversal; identical frames in one
    batch give identical outputs; 384 x 384 outputs (DeiT-base-384) take the same path."""
    rng = np.random.default_rng(2)
    f = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)
    T = D.DeviceTransform(224, device=DEV)
    prm = D.eval_crop_params(300, 400)
    out = T([f] * 128, [prm + (i % 2 == 1,) for i in range(128)])
    assert torch.equal(out[0], out[2]) and torch.equal(out[1], out[127])
    assert torch.equal(out[1], out[0].flip(-1))
    const = np.full((300, 400, 3), 77, dtype=np.uint8)
    o = T([const], [prm]).cpu()
    assert torch.equal(o[0], _ref(const, *prm, 224, False))
    # a frame that already has the output size (the training recipe with RandAugment on the host): both passes are identities
    g = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    assert torch.equal(T([g], [((0, 0, 224, 224), (224, 224), (0, 0), False)]).cpu()[0], O.to_tensor_normalize(g))
    T384 = D.DeviceTransform(384, device=DEV)
    p384 = D.eval_crop_params(500, 700, 384)
    assert torch.equal(T384([f], [((0, 0, 300, 400), (438, 584), (27, 100))]).cpu()[0],
                       _ref(f, (0, 0, 300, 400), (438, 584), (27, 100), 384, False))
    assert p384[1][0] == 438


def test_c_abi_rejects_unplanned_descriptors():
    import ctypes
    from cream_amd import _lib
    lib = _lib.load()
    T = D.DeviceTransform(224, device=DEV)
    descs, nbytes, ws = T.plan([(300, 400)], [D.eval_crop_params(300, 400)])
    pix = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    dd = torch.zeros(ctypes.sizeof(_lib.ImageDesc), dtype=torch.uint8, device=DEV)
    out = torch.empty(1, 3, 224, 224, device=DEV)
    wsb = torch.empty(ws, dtype=torch.uint8, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    call = lambda d, nb, w: lib.cream_image_batch_transform(p(out), p(pix), nb, d, p(dd), 1, 224, 224, T._mean, T._std, p(wsb), w, None)
    assert call(descs, nbytes, ws) == 0
    assert call(descs, nbytes, ws - 16) == -1                      # workspace too small
    assert call(descs, nbytes - 4, ws) == -1                       # frame reaches past the pixel buffer
    descs[0].nrows += 1
    assert call(descs, nbytes, ws) == -1                           # not the plan of these descriptors
    torch.cuda.synchronize()


def test_random_erasing_box_is_noise_and_the_rest_is_untouched():
    """RandomErasing (mode 'pixel') on the normalised output: inside the box the device's counter-based standard-normal noise (the numpy
    restatement within float-function rounding; mean / variance of a large box), outside bit for bit the un-erased result; the box is in
    OUTPUT coordinates (after the mirror)."""
    rng = np.random.default_rng(4)
    f = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)
    T = D.DeviceTransform(224, device=DEV)
    base = D.eval_crop_params(300, 400)
    box = (30, 50, 120, 101, 0xC0FFEE)
    plain = T([f, f], [base + (False, None), base + (True, None)]).cpu()
    erased = T([f, f], [base + (False, box), base + (True, box)]).cpu()
    noise = torch.from_numpy(D.erase_noise_reference(box[4], 3, 224, 224))
    t, l, h, w, _ = box
    for b in range(2):
        inside = torch.zeros(224, 224, dtype=torch.bool)
        inside[t:t + h, l:l + w] = True
        assert torch.equal(erased[b][:, ~inside], plain[b][:, ~inside])
        assert float((erased[b][:, inside] - noise[:, inside]).abs().max()) < 1e-4
    z = erased[0][:, t:t + h, l:l + w]
    assert abs(float(z.mean())) < 0.02 and abs(float(z.var()) - 1.0) < 0.03 and float(z.abs().max()) < 6.5
    assert not torch.equal(erased[0][0, t:t + h, l:l + w], erased[0][1, t:t + h, l:l + w])       # channels differ


def test_evaluate_runs_from_decoded_frames_through_the_device_transform():
    """engine.evaluate fed by DeviceBatches (decoded frames -> device transform) gives the statistics of the same sub-network on the
    tensors the reference's eval transform (Pillow + torch on the host: the oracle's restatement) produces."""
    from cream_amd.autoformer import engine
    from cream_amd.autoformer.supernet import Vision_TransformerSuper
    torch.manual_seed(0)
    model = Vision_TransformerSuper(img_size=224, patch_size=16, embed_dim=256, depth=2, num_heads=4, mlp_ratio=4.0, qkv_bias=True,
                                    num_classes=10, gp=True, relative_position=True, change_qkv=True, max_relative_position=14).to(DEV)
    cfg = dict(layer_num=2, embed_dim=[192] * 2, num_heads=[3] * 2, mlp_ratio=[3.5] * 2)
    rng = np.random.default_rng(9)
    shapes = [(300, 400), (400, 300), (256, 256), (500, 333)]
    loader = [([rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes], [1, 2, 3, 4]) for _ in range(2)]
    got = engine.evaluate(D.DeviceBatches(loader, D.DeviceTransform(224, device=DEV), "eval"), model, amp_dtype=torch.float32,
                          mode="retrain", retrain_config=cfg)
    host = []
    for frames, labels in loader:
        x = torch.stack([O.to_tensor_normalize(O.resized_window(f, *D.eval_crop_params(*f.shape[:2]), (224, 224))) for f in frames])
        host.append((x.to(DEV), torch.tensor(labels, device=DEV)))
    want = engine.evaluate(host, model, amp_dtype=torch.float32, mode="retrain", retrain_config=cfg)
    assert got["loss"] == want["loss"] and got["acc1"] == want["acc1"] and got["acc5"] == want["acc5"]


def test_fuzz_ragged_layouts_against_the_oracle():
    """Random frames (1 x 1 up to 90 x 130), random boxes, resized sizes and windows, frames at ARBITRARY byte offsets with padded row
    strides, straight through the C ABI — every output bit for bit the oracle's."""
    import ctypes
    from cream_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(123)
    for out in (4, 16, 32):
        B = 24
        descs = (_lib.ImageDesc * B)()
        frames, chunks, off = [], [], 0
        for d in descs:
            h, w = int(rng.integers(1, 91)), int(rng.integers(1, 131))
            stride = 3 * w + int(rng.integers(0, 8))
            f = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            pad = int(rng.integers(0, 5))                          # frames start at any byte
            buf = np.zeros(pad + h * stride, dtype=np.uint8)
            for r in range(h):
                buf[pad + r * stride: pad + r * stride + 3 * w] = f[r].reshape(-1)
            bh, bw = int(rng.integers(1, h + 1)), int(rng.integers(1, w + 1))
            bt, bl = int(rng.integers(0, h - bh + 1)), int(rng.integers(0, w - bw + 1))
            rh, rw = out + int(rng.integers(0, 20)), out + int(rng.integers(0, 20))
            d.offset, d.height, d.width, d.row_stride = off + pad, h, w, stride
            d.box_top, d.box_left, d.box_h, d.box_w = bt, bl, bh, bw
            d.resized_h, d.resized_w = rh, rw
            d.win_top, d.win_left = int(rng.integers(0, rh - out + 1)), int(rng.integers(0, rw - out + 1))
            d.flip = int(rng.integers(0, 2))
            frames.append(f)
            chunks.append(buf)
            off += buf.size
        packed = np.concatenate(chunks + [np.zeros((-off) % 4, dtype=np.uint8)])
        ws = lib.cream_image_batch_plan(descs, B, out, out)
        assert ws > 0
        pix = torch.from_numpy(packed).to(DEV)
        dd = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(DEV)
        wsb = torch.empty(ws, dtype=torch.uint8, device=DEV)
        o = torch.empty(B, 3, out, out, device=DEV)
        mean = (ctypes.c_float * 3)(*O.IMAGENET_DEFAULT_MEAN)
        std = (ctypes.c_float * 3)(*O.IMAGENET_DEFAULT_STD)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        assert lib.cream_image_batch_transform(p(o), p(pix), packed.size, descs, p(dd), B, out, out, mean, std, p(wsb), ws, None) == 0
        o = o.cpu()
        for b, (d, f) in enumerate(zip(descs, frames)):
            want = _ref(f, (d.box_top, d.box_left, d.box_h, d.box_w), (d.resized_h, d.resized_w), (d.win_top, d.win_left), out, bool(d.flip))
            assert torch.equal(o[b], want), (out, b)


def test_tinyclip_pipelines_end_to_end_on_the_device():
    """open_clip's image_transform (TinyCLIP/src/open_clip/transform.py:71-122), train and val, with CLIP's statistics: bit for bit
    what Pillow's resize + torch's float ops give on the host."""
    from cream_amd.tinyclip import transform as CT
    rng = np.random.default_rng(21)
    g = torch.Generator().manual_seed(1)
    shapes = [(375, 500), (640, 480), (224, 224), (180, 320)]
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    T = CT.device_transform(224, device=DEV)
    for params in ([CT.train_crop_params(h, w, generator=g) for h, w in shapes], [CT.val_crop_params(h, w) for h, w in shapes],
                   [CT.val_crop_params(h, w, keep_ratio=False) for h, w in shapes]):
        out = T(frames, params).cpu()
        for o, f, (box, resized, window, flip) in zip(out, frames, params):
            want = O.to_tensor_normalize(O.resized_window(f, box, resized, window, (224, 224), flip), CT.OPENAI_DATASET_MEAN, CT.OPENAI_DATASET_STD)
            assert torch.equal(o, want)


def test_image_folder_to_device_batches(tmp_path):
    """Files on disk -> ImageFolderFrames -> frame_loader -> DeviceBatches: the batches are what the reference's
    DataLoader(ImageFolder(root, transform=build_transform(False, args))) yields, bit for bit (Pillow decode + resize on the host
    side of the comparison)."""
    from PIL import Image
    rng = np.random.default_rng(1)
    for cls in ("a", "b"):
        (tmp_path / cls).mkdir()
        for i in range(3):
            arr = rng.integers(0, 256, (int(rng.integers(240, 400)), int(rng.integers(240, 400)), 3), dtype=np.uint8)
            Image.fromarray(arr).save(tmp_path / cls / f"{i}.png")
    ds = D.ImageFolderFrames(str(tmp_path))
    got = list(D.DeviceBatches(D.frame_loader(ds, batch_size=4), D.DeviceTransform(224, device=DEV), "eval"))
    assert [tuple(x.shape) for x, _ in got] == [(4, 3, 224, 224), (2, 3, 224, 224)]
    assert torch.cat([y for _, y in got]).tolist() == [0, 0, 0, 1, 1, 1]
    k = 0
    for x, _ in got:
        for o in x.cpu():
            im = Image.open(ds.samples[k][0]).convert("RGB")
            w, h = im.size
            box, resized, window = D.eval_crop_params(h, w)
            im = im.resize((resized[1], resized[0]), Image.BICUBIC).crop((window[1], window[0], window[1] + 224, window[0] + 224))
            assert torch.equal(o, O.to_tensor_normalize(np.asarray(im)))
            k += 1


@pytest.mark.parametrize("out,width", [(224, 2000), (384, 2500), (224, 3000)])
def test_very_wide_frames_take_the_narrow_row_groups(out, width):
    """Wide boxes with large tables leave LDS for only two rows / one row per step of the horizontal pass (R = 2 / 1): same bytes."""
    rng = np.random.default_rng(width)
    f = rng.integers(0, 256, (40, width, 3), dtype=np.uint8)
    prm = ((0, 0, 40, width), (out, out), (0, 0), True)
    o = D.DeviceTransform(out, device=DEV)([f], [prm]).cpu()[0]
    assert torch.equal(o, _ref(f, *prm[:3], out, True))


def test_device_batches_train_mode_matches_the_same_draws_on_the_host():
    """DeviceBatches in training mode: crop, flip and RandomErasing parameters drawn per image in the recipe's order from one
    random.Random; the same draws replayed on the host give the same tensors outside the erased boxes (bit for bit) and noise inside."""
    rng = np.random.default_rng(6)
    shapes = [(300, 400), (260, 260), (500, 333), (240, 320)] * 4
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    loader = [(frames, list(range(16)))]
    T = D.DeviceTransform(224, device=DEV)
    (x, y), = list(D.DeviceBatches(loader, T, "train", rng=random.Random(11), reprob=0.5))
    assert y.tolist() == list(range(16)) and tuple(x.shape) == (16, 3, 224, 224)
    replay = D.DeviceBatches(loader, T, "train", rng=random.Random(11), reprob=0.5).params_for(shapes)
    erased = 0
    for o, f, (box, resized, window, flip, erase) in zip(x.cpu(), frames, replay):
        want = _ref(f, box, resized, window, 224, flip)
        keep = torch.ones(224, 224, dtype=torch.bool)
        if erase is not None:
            t, l, h, w, _ = erase
            keep[t:t + h, l:l + w] = False
            erased += 1
            assert not torch.equal(o[:, ~keep], want[:, ~keep])
        assert torch.equal(o[:, keep], want[:, keep])
    assert 2 <= erased <= 14
