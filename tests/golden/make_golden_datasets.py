"""Generate tests/golden/datasets_cub.json from the REFERENCE's own dataset code (runs only in the build container).

SURVEY.md §8(f) row 3.  StackGAN_v2/datasets.py is imported from /root/reference with in-memory stand-ins for the absent
third-party names, as make_golden.py does (easydict.EasyDict for miscc/config.py; `torchvision.transforms`, whose classes
are constructed in BaseDataset.__init__ but NOT restated here: the stand-ins raise when called).  What runs is the part of
the reference that needs no torchvision:

  * BirdsDataset.__init__ / load_bbox (pandas) on the synthetic CUB tree of tests/helpers.py::make_cub_tree;
  * BirdsDataset.__getitem__ (train split) with TREE.BRANCH_NUM = 1, transform = None and `norm` replaced by "keep the PIL
    image": the reference's own get_imgs then performs the bounding-box crop (datasets.py:43-52) and returns the cropped
    image untouched; the `random` draws (embedding of the ten, wrong image of another class) are the reference's;
  * the test split's item.

Stored: every bounding box as load_bbox returns it, and per item and seed the image path, label, embedding row index,
wrong-image path, and size + SHA-1 of the cropped real / wrong images.  The torchvision half of the pipeline (Resize /
RandomCrop / RandomHorizontalFlip / ToTensor / Normalize) stays "parity unpinned" against the reference.

Usage:  python tests/golden/make_golden_datasets.py
"""
import hashlib
import json
import os
import random
import sys
import tempfile
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import make_cub_tree  # noqa: E402

REF = '/root/reference/StackGAN_v2'
SEEDS = (0, 1, 2)


def import_reference_datasets():
    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)
        __setattr__ = __setitem__

    class _Absent(object):
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            raise RuntimeError("torchvision is absent: this transform is not part of the pinned path")

    ed = types.ModuleType('easydict'); ed.EasyDict = EasyDict
    tv = types.ModuleType('torchvision'); tvt = types.ModuleType('torchvision.transforms')
    for name in ('Compose', 'ToTensor', 'Normalize', 'Resize', 'RandomCrop', 'RandomHorizontalFlip', 'Scale'):
        setattr(tvt, name, type(name, (_Absent,), {}))
    tv.transforms = tvt
    for name, mod in (('easydict', ed), ('torchvision', tv), ('torchvision.transforms', tvt)):
        sys.modules.setdefault(name, mod)
    sys.path.insert(0, REF)
    import miscc.config as rcfg
    import datasets as rdata
    return rcfg.cfg, rdata


def digest(img):
    a = np.asarray(img, dtype=np.uint8)
    return {"size": [int(img.size[0]), int(img.size[1])], "sha1": hashlib.sha1(a.tobytes()).hexdigest()}


def main():
    cfg, rdata = import_reference_datasets()
    cfg.TREE.BRANCH_NUM = 1
    out = {"seeds": list(SEEDS)}
    with tempfile.TemporaryDirectory() as root, warnings.catch_warnings():
        warnings.simplefilter("ignore")   # pandas: delim_whitespace is deprecated
        items = make_cub_tree(root)
        ds = rdata.BirdsDataset(root, train=True, base_size=64, transform=None)
        ds.norm = lambda img: img
        out["bbox"] = {k: [int(v) for v in box] for k, box in ds.bbox.items()}
        out["len"] = len(ds)
        import pickle
        with open(os.path.join(root, "train", "audio_features_image.pickle"), "rb") as fp:
            emb = pickle.load(fp)   # our own file (make_cub_tree wrote it)
        train = []
        for seed in SEEDS:
            random.seed(seed)
            for idx in range(len(ds)):
                real, wrong, e, path, label = ds[idx]
                row = [k for k in range(10) if np.array_equal(e, emb[idx][k])]
                assert len(real) == 1 and len(wrong) == 1 and len(row) == 1
                # which image was drawn as the wrong one: the only one whose crop matches is found by the test through
                # the digest; its path is recovered here by replaying the reference's draw
                train.append({"seed": seed, "index": idx, "path": path, "label": int(label), "emb_row": row[0],
                              "real": digest(real[0]), "wrong": digest(wrong[0])})
        out["train_items"] = train
        ts = rdata.BirdsDataset(root, train=False, base_size=64, transform=None)
        ts.norm = lambda img: img
        test = []
        for idx in range(len(ts)):
            real, e, path = ts[idx]
            assert np.asarray(e).shape == (10, emb.shape[2])
            test.append({"index": idx, "path": path, "real": digest(real[0])})
        out["test_items"] = test
        # the wrong-image draw on its own (datasets.py:456-462), 40 draws from one seed
        random.seed(11)
        out["wrong_draws"] = [ds.find_wrong_image(ds._get_class(ds.json_data[i % len(ds)])) for i in range(40)]
        assert len(items) == len(ds)
    with open(os.path.join(HERE, "datasets_cub.json"), "w") as fp:
        json.dump(out, fp, indent=1, sort_keys=True)
    print("wrote datasets_cub.json: %d boxes, %d train items, %d test items" % (len(out["bbox"]), len(train), len(test)))


if __name__ == "__main__":
    main()
