"""CPU: the oracle of the input transform (oracle/image_transform_oracle.py: Pillow's 8-bit bicubic resize restated in numpy) against
the committed Pillow-made fixtures and — where Pillow is installed — against Pillow itself; the host-side parameter draws; the plan
entry point of the C ABI (host only, no device work)."""
import random

import numpy as np
import pytest

from oracle import image_transform_oracle as O
from cream_amd.autoformer import data as D

GOLD = np.load(__file__.rsplit("/", 1)[0] + "/golden/image_transform.npz")
NCASES = sum(1 for k in GOLD.files if k.startswith("frame"))


def _case(i):
    p = [int(v) for v in GOLD[f"params{i}"]]
    return GOLD[f"frame{i}"], tuple(p[0:4]), tuple(p[4:6]), tuple(p[6:8]), bool(p[8]), p[9], GOLD[f"u8_{i}"]


@pytest.mark.parametrize("i", range(NCASES))
def test_oracle_matches_pillow_made_fixtures_byte_for_byte(i):
    frame, box, resized, window, flip, size, want = _case(i)
    got = O.resized_window(frame, box, resized, window, (size, size), flip)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_oracle_matches_installed_pillow_on_random_resizes():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for h, w, oh, ow in [(37, 53, 64, 64), (90, 120, 32, 48), (64, 64, 64, 31), (75, 50, 25, 50), (21, 19, 80, 77), (300, 200, 33, 47)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(O.resize_bicubic_u8(img, oh, ow), ref), (h, w, oh, ow)


def test_eval_params_are_those_of_resize_256_center_crop_224():
    # 500 x 375 (W x H) frame: shorter side 375 -> 256, longer int(256 * 500 / 375) = 341; centre window
    assert D.eval_crop_params(375, 500) == ((0, 0, 375, 500), (256, 341), (16, 58))
    assert D.eval_crop_params(500, 375) == ((0, 0, 500, 375), (341, 256), (58, 16))
    assert D.eval_crop_params(224, 224) == ((0, 0, 224, 224), (256, 256), (16, 16))
    assert O.eval_params(375, 500) == D.eval_crop_params(375, 500)


def test_train_params_stay_inside_the_frame_and_the_ranges():
    rng = random.Random(0)
    for _ in range(300):
        h, w = rng.randint(20, 600), rng.randint(20, 600)
        box, resized, window, flip = D.train_crop_params(h, w, rng)
        t, l, bh, bw = box
        assert 0 <= t and 0 <= l and t + bh <= h and l + bw <= w and bh > 0 and bw > 0
        assert resized == (224, 224) and window == (0, 0) and flip in (True, False)
    # a frame narrower than the ratio range falls through ten attempts into the clamped central crop
    class Never(random.Random):
        def uniform(self, a, b):
            return b * 50 if a == 0.08 else super().uniform(a, b)
    box, _, _, _ = D.train_crop_params(400, 100, Never(1))
    assert box == ((400 - int(round(100 / (3. / 4.)))) // 2, 0, int(round(100 / (3. / 4.))), 100)


def test_plan_entry_point_validates_and_sizes_without_a_device():
    from cream_amd import _lib
    lib = _lib.load()
    T = D.DeviceTransform(224, device="cpu")
    descs, nbytes, ws = T.plan([(375, 500), (500, 375)], [D.eval_crop_params(375, 500), D.eval_crop_params(500, 375)])
    assert nbytes == 2 * 375 * 500 * 3 and ws > 0 and ws % 16 == 0
    # rows the window's vertical pass reads: a sub-range of the frame, and the intermediate of image 1 follows image 0's
    assert 0 <= descs[0].row0 and descs[0].row0 + descs[0].nrows <= 375 and descs[1].tmp_off == (descs[0].nrows * 224 * 3 + 15) // 16 * 16
    bad = (_lib.ImageDesc * 1)()
    bad[0].height, bad[0].width, bad[0].row_stride = 10, 10, 30
    bad[0].box_h, bad[0].box_w, bad[0].resized_h, bad[0].resized_w = 10, 11, 224, 224          # box wider than the frame
    assert lib.cream_image_batch_plan(bad, 1, 224, 224) == -1
    big = (_lib.ImageDesc * 1)()
    big[0].height, big[0].width, big[0].row_stride = 100, 6000, 18000
    big[0].box_h, big[0].box_w, big[0].resized_h, big[0].resized_w = 100, 6000, 224, 224       # shrinks x 27: beyond the table
    assert lib.cream_image_batch_plan(big, 1, 224, 224) == -4
    assert lib.cream_image_batch_plan(descs, 2, 224, 222) == -1                               # out_w % 4


def test_random_erasing_params_gate_and_ranges():
    rng = random.Random(3)
    boxes = [D.random_erasing_params(rng) for _ in range(4000)]
    hit = [b for b in boxes if b is not None]
    assert abs(len(hit) / 4000 - 0.25) < 0.03                       # re_prob 0.25 (supernet_train.py's --reprob default)
    for t, l, h, w, seed in hit:
        assert 0 < h < 224 and 0 < w < 224 and 0 <= t <= 224 - h and 0 <= l <= 224 - w and 0 <= seed < 2 ** 32
        assert 0.02 * 224 * 224 * 0.7 <= h * w <= 224 * 224 / 3 * 1.3
    # the plan entry point rejects a box outside the output
    from cream_amd import _lib
    T = D.DeviceTransform(224, device="cpu")
    with pytest.raises(Exception):
        T.plan([(300, 400)], [D.eval_crop_params(300, 400) + (False, (200, 10, 40, 40, 1))])
    n = D.erase_noise_reference(7, 3, 64, 64)
    assert n.shape == (3, 64, 64) and abs(float(n.mean())) < 0.05 and abs(float(n.var()) - 1) < 0.08


def test_tinyclip_crop_params():
    from cream_amd.tinyclip import transform as CT
    import torch
    assert CT.val_crop_params(375, 500) == ((0, 0, 375, 500), (224, 298), (0, 37), False)
    assert CT.val_crop_params(375, 500, keep_ratio=False) == ((0, 0, 375, 500), (224, 224), (0, 0), False)
    g = torch.Generator().manual_seed(5)
    a = [CT.train_crop_params(375, 500, generator=g) for _ in range(50)]
    g = torch.Generator().manual_seed(5)
    assert a == [CT.train_crop_params(375, 500, generator=g) for _ in range(50)]        # torch's generator replays the crops
    for (t, l, h, w), resized, window, flip in a:
        assert 0 <= t and 0 <= l and t + h <= 375 and l + w <= 500 and resized == (224, 224) and not flip
        assert h * w >= 0.9 * 375 * 500 * 0.97                      # scale (0.9, 1.0)


def test_image_folder_frames_lists_and_decodes_like_image_folder(tmp_path):
    """Class indices from the sorted sub-directories, samples in sorted order, frames decoded through PIL's RGB conversion, ragged
    batches through the loader with the reference's samplers (RASampler over it: three repeats per sample)."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    truth = {}
    for cls, names in (("n02", ["b.png", "a.png"]), ("n01", ["x.png"]), ("n03", ["k.PNG", "notes.txt"])):
        (tmp_path / cls).mkdir()
        for n in names:
            if n.endswith("txt"):
                (tmp_path / cls / n).write_text("not an image")
                continue
            arr = rng.integers(0, 256, (int(rng.integers(8, 20)), int(rng.integers(8, 20)), 3), dtype=np.uint8)
            Image.fromarray(arr).save(tmp_path / cls / n)          # PNG: lossless
            truth[(cls, n)] = arr
    ds = D.ImageFolderFrames(str(tmp_path))
    assert ds.classes == ["n01", "n02", "n03"] and ds.class_to_idx == {"n01": 0, "n02": 1, "n03": 2}
    assert [(p.rsplit("/", 2)[1], p.rsplit("/", 1)[1], t) for p, t in ds.samples] == \
        [("n01", "x.png", 0), ("n02", "a.png", 1), ("n02", "b.png", 1), ("n03", "k.PNG", 2)]
    frame, target = ds[2]
    assert target == 1 and np.array_equal(frame, truth[("n02", "b.png")])
    batches = list(D.frame_loader(ds, batch_size=3))
    assert [len(b[0]) for b in batches] == [3, 1] and batches[0][1] == [0, 1, 1] and batches[1][1] == [2]
    sampler = D.RASampler(ds, num_replicas=1, rank=0, shuffle=False)
    assert len(list(iter(sampler))) == sampler.num_selected_samples
