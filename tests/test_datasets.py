"""on-disk formats at the edges of the step (reference StackGAN_v2/datasets.py:420-642,
Audio_to_Image/extract_audio_feature.py:88-96, main.py:126-181).

The reference's transforms need torchvision (absent here).  The torchvision-free half -- load_bbox, the bounding-box crop,
the item tuples and the `random` draws -- is pinned on a fixture the reference's own BirdsDataset produced
(tests/golden/datasets_cub.json, make_golden_datasets.py); the resize / random-crop / flip / normalise half is pinned on its
deterministic arithmetic only and stays "parity unpinned" against the reference.
"""
import json
import os
import pickle
import random

import numpy as np
import pytest
import torch
from PIL import Image

from helpers import CASES, build_nets, configure

from speech_to_image_translation_without_text_amd import datasets as D
from speech_to_image_translation_without_text_amd.miscc.config import cfg


def _make_tree(root, n=12, birds=True, dim=32):
    rng = np.random.RandomState(0)
    img_root = os.path.join(root, "images")
    items, boxes, names = [], [], []
    for i in range(n):
        cls = "%03d.Bird_%d" % (i % 3 + 1, i % 3) if birds else str(i % 3)
        rel = "%s/img_%d.png" % (cls, i)
        os.makedirs(os.path.join(img_root, os.path.dirname(rel)), exist_ok=True)
        w, h = 90 + 7 * i, 120 - 3 * i
        Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(img_root, rel))
        key = "image" if birds else "img"
        items.append({key: rel, "class": cls, "audio": ["a_%d_%d.wav" % (i, k) for k in range(10)], "text": ["t"] * 10})
        boxes.append((i + 1, 10.0 + i, 20.0, 50.0 + i, 40.0))
        names.append((i + 1, rel))
    emb = None
    for split in ("train", "test"):
        with open(os.path.join(root, split + ".json"), "w") as fp:
            json.dump({"image_base_path": img_root, "audio_base_path": os.path.join(root, "audio"), "data": items}, fp)
        emb = rng.randn(n, 10, dim).astype(np.float32)
        os.makedirs(os.path.join(root, split), exist_ok=True)
        with open(os.path.join(root, split, "audio_features_image.pickle"), "wb") as fp:
            pickle.dump(emb, fp)  # exactly what extract_audio_feature.py:93-94 does
    os.makedirs(os.path.join(root, "CUB_200_2011"), exist_ok=True)
    with open(os.path.join(root, "CUB_200_2011", "bounding_boxes.txt"), "w") as fp:
        for b in boxes:
            fp.write("%d %.1f %.1f %.1f %.1f\n" % b)
    with open(os.path.join(root, "CUB_200_2011", "images.txt"), "w") as fp:
        for nm in names:
            fp.write("%d %s\n" % nm)
    return emb  # the test split's array (written last); the train split's is re-read by the tests that need it


def test_embedding_pickle_roundtrip_and_refusal(tmp_path):
    arr = np.random.RandomState(1).randn(5, 10, 16).astype(np.float32)
    p = str(tmp_path / "train" / "audio_features_x.pickle")
    D.save_embedding_pickle(arr, p)
    with open(p, "rb") as fp:
        assert np.array_equal(pickle.load(fp), arr)       # plain ndarray pickle, as the reference's reader expects
    back = D.load_embedding_pickle(p)
    assert back.dtype == np.float32 and np.array_equal(back, arr)
    for proto in (2, 3, 4, 5):                             # files written by other Python / pickle versions
        q = str(tmp_path / ("p%d.pickle" % proto))
        with open(q, "wb") as fp:
            pickle.dump(arr, fp, protocol=proto)
        assert np.array_equal(D.load_embedding_pickle(q), arr)
    evil = str(tmp_path / "evil.pickle")
    with open(evil, "wb") as fp:
        pickle.dump(os.path.join, fp)                      # any non-numpy global must be refused, not resolved
    with pytest.raises(pickle.UnpicklingError):
        D.load_embedding_pickle(evil)
    obj = str(tmp_path / "obj.pickle")
    with open(obj, "wb") as fp:
        pickle.dump({"a": 1}, fp)
    with pytest.raises(pickle.UnpicklingError):
        D.load_embedding_pickle(obj)


def test_transform_arithmetic():
    # torchvision.transforms.Resize(int): short side to size, long side int(size * long / short)
    assert D.Resize(76).output_size(90, 120) == (76, int(76 * 120 / 90))
    assert D.Resize(76).output_size(120, 90) == (int(76 * 120 / 90), 76)
    assert D.Resize(64).output_size(64, 100) == (64, 100)
    # datasets.py:43-52: r = int(max(w, h) * 0.75), centre = (int((2x + w) / 2), int((2y + h) / 2))
    assert D.crop_box([10, 20, 50, 40], 200, 100) == (0, 3, 72, 77)
    assert D.crop_box([0, 0, 4, 4], 30, 30) == (0, 0, 12, 12)              # r is at least 10
    assert D.crop_box([150, 60, 100, 80], 200, 100) == (125, 25, 200, 100)  # clipped to the image
    img = Image.fromarray(np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3))
    t = D.to_normalized_tensor(img)
    a = np.asarray(img).astype(np.float32)
    ref = ((torch.from_numpy(a).permute(2, 0, 1) / 255) - 0.5) / 0.5
    assert t.shape == (3, 4, 6) and torch.equal(t, ref)
    assert torch.equal(D.to_uint8_hwc(img), torch.from_numpy(np.asarray(img)))
    tr = D.default_image_transform(256)
    assert [type(x).__name__ for x in tr.transforms] == ["Resize", "RandomCrop", "RandomHorizontalFlip"]
    assert tr.transforms[0].size == 304 and tr.transforms[1].size == 256


@pytest.mark.parametrize("birds", [True, False])
def test_dataset_items_follow_reference_layout(tmp_path, birds):
    configure(CASES['full3_fwd'])
    _make_tree(str(tmp_path), birds=birds)
    emb = D.load_embedding_pickle(str(tmp_path / "train" / "audio_features_image.pickle"))
    cls = D.BirdsDataset if birds else D.FlowersDataset
    random.seed(3)
    ds = cls(str(tmp_path), train=True, base_size=cfg.TREE.BASE_SIZE, transform=D.default_image_transform(256))
    assert len(ds) == 12 and ds.imsize == [64, 128, 256]
    for idx in (0, 5, 11):
        real, wrong, e, path, label = ds[idx]
        assert [tuple(t.shape) for t in real] == [(3, 64, 64), (3, 128, 128), (3, 256, 256)]
        assert [tuple(t.shape) for t in wrong] == [(3, 64, 64), (3, 128, 128), (3, 256, 256)]
        assert all(t.dtype == torch.float32 and float(t.min()) >= -1 and float(t.max()) <= 1 for t in real + wrong)
        assert any(np.array_equal(e, emb[idx][k]) for k in range(10))    # one of the ten spoken captions
        assert label == idx % 3 + (1 if birds else 0)
        assert path == ds._get_img(ds.json_data[idx])
    # the pyramid: smaller branches are PIL bilinear resizes of the largest one (datasets.py:57-64)
    real, _, _, _, _ = ds[2]
    big = Image.fromarray((((real[2] * 0.5 + 0.5) * 255).round().clamp(0, 255).byte()).permute(1, 2, 0).numpy())
    assert torch.equal(real[1], D.to_normalized_tensor(big.resize((128, 128), Image.BILINEAR)))
    assert torch.equal(real[0], D.to_normalized_tensor(big.resize((64, 64), Image.BILINEAR)))
    # test split: all ten embeddings, no wrong image (datasets.py:489-499)
    ts = cls(str(tmp_path), train=False, base_size=64, transform=D.default_image_transform(256))
    real, e, path = ts[4]
    assert e.shape == (10, 32) and len(real) == 3
    # wrong images come from another class
    for _ in range(20):
        wp = ds.find_wrong_image(ds._get_class(ds.json_data[0]))
        other = [it for it in ds.json_data if ds._get_img(it) == wp][0]
        assert ds._get_class(other) != ds._get_class(ds.json_data[0])
    if birds:
        assert ds.bbox["001.Bird_0/img_0"] == [10, 20, 50, 40]


def test_dataloader_batches_and_rank_sharding(tmp_path):
    configure(CASES['full3_fwd'])
    _make_tree(str(tmp_path), birds=True)
    ds = D.BirdsDataset(str(tmp_path), train=True, transform=D.default_image_transform(256), device_normalize=True)
    dl = D.make_dataloader(ds, 4, shuffle=False)
    real, wrong, emb, paths, labels = next(iter(dl))
    assert [tuple(t.shape) for t in real] == [(4, 64, 64, 3), (4, 128, 128, 3), (4, 256, 256, 3)]
    assert real[0].dtype == torch.uint8 and emb.shape == (4, 32) and emb.dtype == torch.float32
    assert len(paths) == 4 and labels.tolist() == [1, 2, 3, 1]
    seen = []
    for r in range(2):
        dlr = D.make_dataloader(ds, 3, distributed=True, rank=r, world_size=2)
        dlr.sampler.set_epoch(0)
        idx = list(iter(dlr.sampler))
        assert len(idx) == 6
        seen += idx
    assert sorted(seen) == list(range(12))               # disjoint shards that cover the split


@pytest.mark.gpu
def test_device_normalisation_is_bit_identical(gpu):
    from speech_to_image_translation_without_text_amd import ops
    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (3, 40, 24, 3), dtype=torch.uint8, generator=g)
    ref = torch.stack([D.to_normalized_tensor(Image.fromarray(u8[b].numpy())) for b in range(3)])
    out = ops.images_from_uint8_hwc(u8.to(gpu))
    assert torch.equal(out.cpu(), ref)


@pytest.mark.gpu
def test_train_step_from_dataloader_batch(gpu, tmp_path):
    """A DataLoader batch in the reference's tuple layout goes through prepare_data and one full train step; the
    uint8 (device-normalised) and float (reference-style) datasets give the same losses."""
    from speech_to_image_translation_without_text_amd import trainer as T
    case = CASES['small3']
    configure(case)
    _make_tree(str(tmp_path), birds=True, dim=case['t'])
    size = cfg.TREE.BASE_SIZE * 4
    losses = []
    for dn in (False, True):
        random.seed(11)
        netG, netsD = build_nets(case)
        netG.to(gpu)
        [d.to(gpu) for d in netsD]
        tr = T.condGANTrainer(None, None, size, False)
        tr.build(netG, netsD)
        ds = D.BirdsDataset(str(tmp_path), train=True, base_size=cfg.TREE.BASE_SIZE,
                            transform=D.default_image_transform(size), device_normalize=dn)
        batch = next(iter(D.make_dataloader(ds, 8, shuffle=False)))
        _, real, wrong, e, labels = tr.prepare_data(batch)
        assert real[2].shape == (8, 3, size, size) and real[2].dtype == torch.float32
        g = torch.Generator(device=gpu).manual_seed(2)
        noise = torch.randn(8, cfg.GAN.Z_DIM, device=gpu, generator=g)
        eps = torch.randn(8, cfg.GAN.EMBEDDING_DIM, device=gpu, generator=g)
        errD, errG, kl = tr.train_step(real, wrong, e, labels, noise, eps)
        losses.append((float(errD), float(errG), float(kl)))
        assert all(np.isfinite(v) for v in losses[-1])
    assert losses[0] == pytest.approx(losses[1], rel=1e-5)


def test_places_subset_labels_and_paths(tmp_path):
    """datasets.py:567-580: class from the directory name, image path without the leading './'."""
    configure(CASES['full3_fwd'])
    rng = np.random.RandomState(0)
    root = str(tmp_path)
    img_root = os.path.join(root, "images")
    names = ['bedroom', 'kitchenette', 'living_room', 'bedroom']
    items = []
    for i, cls in enumerate(names):
        os.makedirs(os.path.join(img_root, cls), exist_ok=True)
        Image.fromarray(rng.randint(0, 256, (80, 100, 3), dtype=np.uint8)).save(os.path.join(img_root, cls, "%d.png" % i))
        items.append({"image": "./%s/%d.png" % (cls, i), "class": cls, "audio": [], "text": []})
    for split in ("train", "test"):
        with open(os.path.join(root, split + ".json"), "w") as fp:
            json.dump({"image_base_path": img_root, "audio_base_path": "", "data": items}, fp)
        D.save_embedding_pickle(rng.randn(4, 10, 8).astype(np.float32),
                                os.path.join(root, split, "audio_features_cnn_googlenet.pickle"))
    ds = D.PlacesSubSet(root, train=True, transform=D.default_image_transform(256))
    real, wrong, e, path, label = ds[1]
    assert path == "kitchenette/1.png" and label == 5 and e.shape == (8,) and real[2].shape == (3, 256, 256)
    assert [ds._get_class(it) for it in ds.json_data] == [0, 5, 6, 0]


def test_birds_dataset_against_the_reference_fixture(tmp_path):
    """tests/golden/datasets_cub.json holds what the REFERENCE's BirdsDataset returned on the tree of
    helpers.make_cub_tree (make_golden_datasets.py: load_bbox, the bounding-box crop inside its get_imgs, the item tuples
    and the `random` draws; datasets.py:40-52, 424-433, 456-499, 504-564).  Same tree, same seeds, this package's dataset:
    every box, path, label, embedding row and cropped image must be identical.  The torchvision transforms are not part of
    the fixture (absent library): transform = None and the crop is compared as the PIL image it is."""
    import hashlib
    from helpers import make_cub_tree
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "datasets_cub.json")) as fp:
        gold = json.load(fp)
    saved = cfg.TREE.BRANCH_NUM
    cfg.TREE.BRANCH_NUM = 1
    try:
        make_cub_tree(str(tmp_path))
        emb = D.load_embedding_pickle(str(tmp_path / "train" / "audio_features_image.pickle"))

        def digest(img):
            return {"size": [img.size[0], img.size[1]],
                    "sha1": hashlib.sha1(np.asarray(img, dtype=np.uint8).tobytes()).hexdigest()}

        ds = D.BirdsDataset(str(tmp_path), train=True, base_size=64, transform=None)
        ds.norm = lambda img: img
        assert len(ds) == gold["len"]
        assert ds.bbox == gold["bbox"]
        it = iter(gold["train_items"])
        for seed in gold["seeds"]:
            random.seed(seed)
            for idx in range(len(ds)):
                g = next(it)
                real, wrong, e, path, label = ds[idx]
                assert (g["seed"], g["index"]) == (seed, idx)
                assert path == g["path"] and label == g["label"]
                assert np.array_equal(e, emb[idx][g["emb_row"]])
                assert digest(real[0]) == g["real"], (seed, idx)
                assert digest(wrong[0]) == g["wrong"], (seed, idx)
        ts = D.BirdsDataset(str(tmp_path), train=False, base_size=64, transform=None)
        ts.norm = lambda img: img
        for g in gold["test_items"]:
            real, e, path = ts[g["index"]]
            assert path == g["path"] and e.shape[0] == 10 and digest(real[0]) == g["real"]
        random.seed(11)
        draws = [ds.find_wrong_image(ds._get_class(ds.json_data[i % len(ds)])) for i in range(40)]
        assert draws == gold["wrong_draws"]
        # the crop rule alone on the fixture's boxes: sizes follow crop_box (datasets.py:43-52)
        for g in (x for x in gold["train_items"] if x["seed"] == gold["seeds"][0]):
            w, h = Image.open(os.path.join(ds.image_folder, g["path"])).size
            x1, y1, x2, y2 = D.crop_box(gold["bbox"][g["path"][:-4]], w, h)
            assert [x2 - x1, y2 - y1] == g["real"]["size"]
    finally:
        cfg.TREE.BRANCH_NUM = saved
