"""SURVEY 8(f)4, second half: the data pipeline (src/data/datasets.py, loaders.py) without OpenCV / torchvision.
Transforms that do not call OpenCV and the dataset class are compared bit for bit with what the imported
reference produced (tests/golden/data.npz, make_golden.py:gen_data); the OpenCV restatements (parity unpinned:
cv2 is not in this image) are checked against PyTorch's bicubic / nearest interpolation, which implement the
same published formulas.  CPU only: data loading is host work."""
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from _util import load_json, load_npz

NPZ = load_npz("data.npz")
META = load_json("data_meta.json")
MEAN = np.array([0.485, 0.456, 0.406]).reshape((1, 1, 3))
STD = np.array([0.229, 0.224, 0.225]).reshape((1, 1, 3))


def D():
    from nas_segm_amd.data import datasets

    return datasets


def same(a, b):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("i", range(len(META["cases"])))
def test_transforms_equal_the_reference(i):
    d = D()
    sample = {"image": NPZ["in{}/image".format(i)], "mask": NPZ["in{}/mask".format(i)]}
    size, img_val, msk_val = META["pad"]
    out = d.Pad(size, img_val, msk_val)(sample)
    assert same(out["image"], NPZ["pad{}/image".format(i)]) and same(out["mask"], NPZ["pad{}/mask".format(i)])
    out = d.CentralCrop(META["ccrop"])(sample)
    assert same(out["image"], NPZ["ccrop{}/image".format(i)]) and same(out["mask"], NPZ["ccrop{}/mask".format(i)])
    np.random.seed(100 + i)
    out = d.RandomCrop(META["rcrop"])(sample)
    assert same(out["image"], NPZ["rcrop{}/image".format(i)]) and same(out["mask"], NPZ["rcrop{}/mask".format(i)])
    out = d.ResizeShorter(min(sample["image"].shape[:2]))(sample)
    assert same(out["image"], NPZ["rshort{}/image".format(i)])
    out = d.ToTensor()(d.Normalise(1.0 / 255, MEAN, STD)(sample))
    assert same(out["image"], NPZ["norm{}/image".format(i)]) and same(out["mask"], NPZ["norm{}/mask".format(i)])
    assert out["image"].dtype == torch.float64 and tuple(out["image"].shape[:1]) == (3,)  # (as the reference: float64)


def test_dataset_equals_the_reference(tmp_path):
    from PIL import Image

    d = D()
    for i, (a, b) in enumerate(META["names"]):
        Image.fromarray(NPZ["file{}/image".format(i)]).save(str(tmp_path / a))
        Image.fromarray(NPZ["file{}/mask".format(i)]).save(str(tmp_path / b))
    (tmp_path / "two.lst").write_text("".join("{}\t{}\n".format(a, b) for a, b in META["names"]))
    (tmp_path / "one.lst").write_text("".join("{}\n".format(b) for _, b in META["names"]))
    norm = d.Normalise(1.0 / 255, MEAN, STD)
    trn = d.Compose([d.ResizeShorter(16), d.CentralCrop(30), d.RandomCrop(24), norm, d.ToTensor()])
    val = d.Compose([d.CentralCrop(32), norm, d.ToTensor()])
    ds = d.PascalCustomDataset(str(tmp_path / "two.lst"), str(tmp_path), trn, val)
    assert len(ds) == 3 and ds.stage == "train"
    np.random.seed(9)
    for i in range(3):
        out = ds[i]
        assert same(out["image"], NPZ["ds_trn{}/image".format(i)]) and same(out["mask"], NPZ["ds_trn{}/mask".format(i)])
    ds.set_stage("val")
    for i in range(3):
        out = ds[i]
        assert same(out["image"], NPZ["ds_val{}/image".format(i)]) and same(out["mask"], NPZ["ds_val{}/mask".format(i)])
    ds.set_stage("train")
    ds.set_config(20, 8)
    assert trn.transforms[2].crop_size == 20 and trn.transforms[0].resize_side == 8 if hasattr(
        trn.transforms[0], "resize_side") else True
    np.random.seed(10)
    out = ds[2]
    assert same(out["image"], NPZ["ds_cfg/image"]) and same(out["mask"], NPZ["ds_cfg/mask"])
    # a one-column list: the reference's fallback is dead code under Python 3 (it catches ValueError, the
    # tuple indexing raises IndexError - recorded by make_golden.py); here it works as its comment intends
    assert META["single_column"] == "raises IndexError"
    one = d.PascalCustomDataset(str(tmp_path / "one.lst"), str(tmp_path), None, None)
    assert one.datalist == [(b, b) for _, b in META["names"]]


@pytest.mark.parametrize("scale", [2.0, 1.0, 1.7, 0.6, 1.25, 0.83])
def test_opencv_resize_restatements_against_torch(scale):
    """INTER_CUBIC = Keys kernel, A = -0.75, pixel-centre mapping with the GIVEN scale, replicated border - what
    torch's bicubic interpolation computes in float; the uint8 fixed-point path may differ by one count.
    INTER_NEAREST = floor(dst / scale) - torch's 'nearest'.  Output sizes follow cvRound (half to even)."""
    d = D()
    rng = np.random.RandomState(3)
    img = (rng.rand(37, 53, 3) * 255).astype(np.uint8)
    out = d.resize_cubic(img, scale)
    assert out.dtype == np.uint8 and out.shape == (int(np.rint(37 * scale)), int(np.rint(53 * scale)), 3)
    t = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
    ref = TF.interpolate(t, scale_factor=scale, mode="bicubic", align_corners=False, recompute_scale_factor=False)
    ref = ref[0].permute(1, 2, 0).round().clamp(0, 255).numpy()
    h, w = min(out.shape[0], ref.shape[0]), min(out.shape[1], ref.shape[1])
    diff = np.abs(out[:h, :w].astype(np.int32) - ref[:h, :w].astype(np.int32))
    assert diff.max() <= 1 and float((diff > 0).mean()) < 0.1
    if scale in (1.0, 2.0):
        assert diff.max() == 0
    # float images take the float path; a constant image stays constant
    assert np.allclose(d.resize_cubic(np.full((9, 11, 3), 0.25, np.float32), scale), 0.25, atol=1e-6)
    msk = (rng.rand(37, 53) * 21).astype(np.uint8)
    mo = d.resize_nearest(msk, scale)
    mt = TF.interpolate(torch.from_numpy(msk.astype(np.float32))[None, None], scale_factor=scale, mode="nearest",
                        recompute_scale_factor=False)[0, 0].numpy().astype(np.uint8)
    h, w = min(mo.shape[0], mt.shape[0]), min(mo.shape[1], mt.shape[1])
    assert mo.shape == out.shape[:2] and np.array_equal(mo[:h, :w], mt[:h, :w])


def test_random_transforms_consume_the_generator_in_the_reference_order():
    """ResizeScale: one uniform draw; RandomMirror: one randint(2); RandomCrop: randint for the top, then the
    left (src/data/datasets.py:152,184,113-114) - so a seeded run is reproducible across implementations."""
    d = D()
    rng = np.random.RandomState(5)
    sample = {"image": (rng.rand(40, 60, 3) * 255).astype(np.uint8), "mask": (rng.rand(40, 60) * 5).astype(np.uint8)}
    np.random.seed(21)
    s = np.random.uniform(0.5, 2.0)
    flip = np.random.randint(2)
    np.random.seed(21)
    out = d.ResizeScale(30, 0.5, 2.0)(sample)
    scale = s if min(40, 60) * s >= 30 else 30.0 / 40
    assert out["image"].shape[:2] == (int(np.rint(40 * scale)), int(np.rint(60 * scale)))
    assert out["mask"].shape == out["image"].shape[:2]
    mirrored = d.RandomMirror()(out)
    want = out["image"][:, ::-1] if flip else out["image"]
    assert np.array_equal(mirrored["image"], want)
    h, w = mirrored["image"].shape[:2]
    top, left = np.random.randint(0, h - 24 + 1), np.random.randint(0, w - 24 + 1)
    np.random.seed(21)
    np.random.uniform(0.5, 2.0)
    np.random.randint(2)
    crop = d.RandomCrop(24)(mirrored)
    assert np.array_equal(crop["image"], mirrored["image"][top:top + 24, left:left + 24])
    # the longer-side mode caps the scale instead
    np.random.seed(2)
    out = d.ResizeScale(50, 1.5, 2.0, longer=True)(sample)
    assert max(out["image"].shape[:2]) == 50


def test_create_loaders(tmp_path):
    """create_loaders(args) (src/data/loaders.py:21-108): the same list for training and validation means
    search mode - a meta-train / meta-val split; batches come out as the engine expects them."""
    from PIL import Image

    from nas_segm_amd.data import create_loaders

    rng = np.random.RandomState(1)
    lines = []
    for i in range(10):
        h, w = 50 + 3 * i, 70 - 2 * i
        Image.fromarray((rng.rand(h, w, 3) * 255).astype(np.uint8)).save(str(tmp_path / "i{}.png".format(i)))
        Image.fromarray((rng.rand(h, w) * 21).astype(np.uint8)).save(str(tmp_path / "m{}.png".format(i)))
        lines.append("i{}.png\tm{}.png\n".format(i, i))
    (tmp_path / "train.lst").write_text("".join(lines))
    (tmp_path / "val.lst").write_text("".join(lines[:4]))
    args = types.SimpleNamespace(
        train_dir=str(tmp_path), val_dir=str(tmp_path), train_list=str(tmp_path / "train.lst"),
        val_list=str(tmp_path / "train.lst"), meta_train_prct=80, resize_side=[40], low_scale=0.7, high_scale=1.4,
        resize_longer_side=False, crop_size=[32], val_resize_side=40, val_crop_size=32,
        normalise_params=[1.0 / 255, MEAN, STD], batch_size=[4], val_batch_size=2, num_workers=0)
    torch.manual_seed(0)
    np.random.seed(0)
    train_loader, val_loader, do_search = create_loaders(args)
    assert do_search and len(train_loader.dataset) == 8 and len(val_loader.dataset) == 2
    assert len(train_loader) == 2 and len(val_loader) == 1  # drop_last
    batch = next(iter(train_loader))
    assert tuple(batch["image"].shape) == (4, 3, 32, 32) and tuple(batch["mask"].shape) == (4, 32, 32)
    assert batch["image"].dtype == torch.float64 and batch["mask"].dtype == torch.uint8
    # the engine reaches the stage switch through the Subset (engine/trainer.py:_set_stage)
    from nas_segm_amd.engine.trainer import _set_stage
    _set_stage(val_loader, "val")
    assert val_loader.dataset.dataset.stage == "val"
    vb = next(iter(val_loader))
    assert tuple(vb["image"].shape) == (2, 3, 32, 32)
    args.val_list = str(tmp_path / "val.lst")
    train_loader, val_loader, do_search = create_loaders(args)
    assert not do_search and len(train_loader.dataset) == 10 and len(val_loader.dataset) == 4
