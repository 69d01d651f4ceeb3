"""On-disk formats at the edges of the train step.

Mirrors the reference's speech-conditioned datasets (`StackGAN_v2/datasets.py:420-642`) and the way `main.py:126-181`
wires them into a DataLoader:

* ``<root>/<split>.json``: ``{"image_base_path", "audio_base_path", "data": [{"image"|"img", "class", "audio",
  "text"}, ...]}`` (datasets.py:425-429);
* ``<root>/<split>/audio_features_<switch>.pickle``: ONE pickled float ndarray ``(N, 10, 1024)`` -- ten spoken
  captions per image -- as `Audio_to_Image/extract_audio_feature.py:88-96` writes it; a training sample takes one of
  the ten at random (datasets.py:470-474, 481), a test sample takes all ten (datasets.py:496);
* images: any PIL-readable file; birds are cropped around ``CUB_200_2011/bounding_boxes.txt`` (datasets.py:43-52,
  527-551); `main.py:127-131`'s transform (resize short side to 76/64 of the final size, random crop, random
  flip), then one image per branch: the full crop for the last branch, `Resize(imsize[i])` of it for the others
  (datasets.py:57-64).

What is different by design:

* the embedding pickle is read by a RESTRICTED unpickler that can only rebuild numpy arrays -- a dataset file can no
  longer execute code on load (the reference calls `pickle.load`);
* with ``device_normalize=True`` samples stay uint8 HWC on the host and the collated batch is normalised on the GPU
  (`ops.images_from_uint8_hwc` -> `s2i_u8_to_image`, bit-identical to ToTensor + Normalize); the host-to-device copy
  is 4x smaller.  The default returns the reference's float CHW tensors.

torchvision is not a dependency: the three PIL transforms are restated here with torchvision's arithmetic.  The
random draws use Python's `random` (as the torchvision of the reference's era does); they cannot be matched draw by
draw against another torchvision version, so this row's parity is pinned on the deterministic parts only (crop box
arithmetic, resize sizes, normalisation, pickle layout) -- see tests/test_datasets.py.
"""
import json
import os
import pickle
import random

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image

from .miscc.config import cfg

IMG_EXTENSIONS = ['.jpg', '.JPG', '.jpeg', '.JPEG', '.png', '.PNG', '.ppm', '.PPM', '.bmp', '.BMP']


def is_image_file(filename):
    return any(filename.endswith(extension) for extension in IMG_EXTENSIONS)


# ---- embedding pickles ---------------------------------------------------------------------------------------------
_ALLOWED_GLOBALS = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
}


def _latin1_encode(text, encoding="latin1"):
    if encoding.lower().replace("-", "") != "latin1":
        raise pickle.UnpicklingError("embedding pickle asks for codec %r" % (encoding,))
    return text.encode("latin1")


class _ArrayOnlyUnpickler(pickle.Unpickler):
    """Rebuilds numpy arrays (and lists/tuples/dicts of them); any other global is refused."""

    def find_class(self, module, name):
        if (module, name) == ("_codecs", "encode"):
            return _latin1_encode  # protocol-2 pickles carry the array bytes as a latin-1 string
        if (module, name) in _ALLOWED_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("embedding pickle refers to %s.%s: only numpy arrays are accepted" % (module, name))


def load_embedding_pickle(path):
    """`audio_features_<switch>.pickle` -> float32 ndarray (N, 10, D).  Object arrays are refused (they would
    unpickle arbitrary payloads element by element)."""
    with open(path, "rb") as fp:
        arr = _ArrayOnlyUnpickler(fp).load()
    if isinstance(arr, (list, tuple)):
        arr = np.stack([np.asarray(a) for a in arr])
    if not isinstance(arr, np.ndarray) or arr.dtype == object:
        raise pickle.UnpicklingError("%s does not hold a numeric ndarray" % path)
    return arr


def save_embedding_pickle(arr, path):
    """The writer side (extract_audio_feature.py:93-96): one `pickle.dump` of the ndarray."""
    arr = np.ascontiguousarray(arr)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as fp:
        pickle.dump(arr, fp)


# ---- PIL transforms (torchvision.transforms arithmetic) ------------------------------------------------------------
class Resize:
    """Short side to `size`, aspect kept, bilinear: torchvision.transforms.Resize(int)."""

    def __init__(self, size):
        self.size = int(size)

    def output_size(self, w, h):
        s = self.size
        if (w <= h and w == s) or (h <= w and h == s):
            return w, h
        if w < h:
            return s, int(s * h / w)
        return int(s * w / h), s

    def __call__(self, img):
        w, h = img.size
        ow, oh = self.output_size(w, h)
        if (ow, oh) == (w, h):
            return img
        return img.resize((ow, oh), Image.BILINEAR)


class RandomCrop:
    def __init__(self, size):
        self.size = int(size)

    def __call__(self, img):
        w, h = img.size
        t = self.size
        if w == t and h == t:
            return img
        i = random.randint(0, h - t)
        j = random.randint(0, w - t)
        return img.crop((j, i, j + t, i + t))


class RandomHorizontalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img):
        if random.random() < self.p:
            return img.transpose(Image.FLIP_LEFT_RIGHT)
        return img


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


def default_image_transform(imsize):
    """main.py:127-131."""
    return Compose([Resize(int(imsize * 76 / 64)), RandomCrop(imsize), RandomHorizontalFlip()])


def to_normalized_tensor(img):
    """ToTensor + Normalize((.5,.5,.5), (.5,.5,.5)) (datasets.py:440-442): float CHW in [-1, 1]."""
    a = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())
    t = a.permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    return t.sub_(0.5).div_(0.5)


def to_uint8_hwc(img):
    """Sample left as uint8 HWC; `ops.images_from_uint8_hwc` normalises the collated batch on the device."""
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())


def crop_box(bbox, width, height):
    """datasets.py:43-52: a square of 1.5x the larger bbox side around the bbox centre, clipped to the image."""
    r = int(np.maximum(bbox[2], bbox[3]) * 0.75)
    r = max(r, 10)
    center_x = int((2 * bbox[0] + bbox[2]) / 2)
    center_y = int((2 * bbox[1] + bbox[3]) / 2)
    y1 = int(np.maximum(0, center_y - r))
    y2 = int(np.minimum(height, center_y + r))
    x1 = int(np.maximum(0, center_x - r))
    x2 = int(np.minimum(width, center_x + r))
    return x1, y1, x2, y2


def get_imgs(img_path, imsize, bbox=None, transform=None, normalize=None):
    """datasets.py:40-66: one tensor per branch, the last branch at the transform's full size."""
    img = Image.open(img_path).convert('RGB')
    width, height = img.size
    if bbox is not None:
        img = img.crop(crop_box(bbox, width, height))
    if transform is not None:
        img = transform(img)
    ret = []
    for i in range(cfg.TREE.BRANCH_NUM):
        re_img = Resize(imsize[i])(img) if i < cfg.TREE.BRANCH_NUM - 1 else img
        ret.append(normalize(re_img))
    return ret


# ---- datasets --------------------------------------------------------------------------------------------------------
class BaseDataset(data.Dataset):
    """datasets.py:420-501.  Train items: (real images per branch, wrong images per branch, embedding (D,),
    image path, class label); test items: (real images per branch, embeddings (10, D), image path)."""

    def __init__(self, data_root, train=True, base_size=64, transform=None, target_transform=None,
                 feature_switch='image', device_normalize=False):
        split = "train" if train else "test"
        self.data_root = data_root
        with open(os.path.join(self.data_root, "{}.json".format(split))) as fp:
            self.json_data_all = json.load(fp)
        self.image_folder = self.json_data_all["image_base_path"]
        self.audio_folder = self.json_data_all['audio_base_path']
        self.json_data = self.json_data_all['data']
        embedding_path = os.path.join(data_root, split, "audio_features_{}.pickle".format(feature_switch))
        print("load features from: {}".format(embedding_path))
        self.embedding = load_embedding_pickle(embedding_path)
        if len(self.embedding) != len(self.json_data):
            raise ValueError("%s holds %d entries for %d images" % (embedding_path, len(self.embedding),
                                                                    len(self.json_data)))
        self.imsize = []
        for _ in range(cfg.TREE.BRANCH_NUM):
            self.imsize.append(base_size)
            base_size = base_size * 2
        self.transform = transform
        self.target_transform = target_transform
        self.norm = to_uint8_hwc if device_normalize else to_normalized_tensor
        self.iterator = self.prepare_train_pairs if train else self.prepare_test_pairs

    def __len__(self):
        return len(self.json_data)

    def _get_img(self, item):
        return item['img']

    def _get_class(self, item):
        return int(item['class'])

    def _get_bbox(self, image_path):
        return None

    def find_wrong_image(self, base_class_label):
        while True:
            json_data, _ = self.get_rand(self.json_data)
            image_path, class_label = self._get_img(json_data), self._get_class(json_data)
            if class_label != base_class_label:
                break
        return image_path

    @staticmethod
    def get_rand(feature):
        rand_idx = random.randint(0, len(feature) - 1)
        return feature[rand_idx], rand_idx

    def _load(self, image_path, bbox=None):
        return get_imgs(os.path.join(self.image_folder, image_path), self.imsize, bbox=bbox,
                        transform=self.transform, normalize=self.norm)

    def prepare_train_pairs(self, index):
        json_data = self.json_data[index]
        image_path, class_label = self._get_img(json_data), self._get_class(json_data)
        embedding, _ = self.get_rand(self.embedding[index])
        wrong_image_path = self.find_wrong_image(class_label)
        real_image = self._load(image_path, self._get_bbox(image_path))
        wrong_image = self._load(wrong_image_path, self._get_bbox(wrong_image_path))
        return real_image, wrong_image, embedding, image_path, class_label

    def prepare_test_pairs(self, index):
        json_data = self.json_data[index]
        image_path = self._get_img(json_data)
        embedding = self.embedding[index]
        real_image = self._load(image_path)  # the reference does not crop test images either (datasets.py:553-561)
        return real_image, embedding, image_path

    def __getitem__(self, index):
        return self.iterator(index)


class BirdsDataset(BaseDataset):
    """datasets.py:504-564: `image` key, class = leading number of "<nnn>.<name>", CUB bounding boxes."""

    def __init__(self, data_root, train=True, base_size=64, transform=None, target_transform=None,
                 feature_switch='image', device_normalize=False):
        super().__init__(data_root, train, base_size, transform, target_transform, feature_switch, device_normalize)
        self.bbox = self.load_bbox()

    def load_bbox(self):
        """`bounding_boxes.txt`: "<id> <x> <y> <w> <h>"; `images.txt`: "<id> <relative path>"; keyed by the path
        without its extension, values truncated to int as the reference's `.astype(int)` does."""
        boxes = []
        with open(os.path.join(self.data_root, 'CUB_200_2011/bounding_boxes.txt')) as fp:
            for line in fp:
                parts = line.split()
                if parts:
                    boxes.append([int(float(v)) for v in parts[1:5]])
        filenames = []
        with open(os.path.join(self.data_root, 'CUB_200_2011/images.txt')) as fp:
            for line in fp:
                parts = line.split()
                if parts:
                    filenames.append(parts[1])
        print('Total filenames: ', len(filenames), filenames[0] if filenames else None)
        return {name[:-4]: box for name, box in zip(filenames, boxes)}

    def _get_img(self, item):
        return item['image']

    def _get_class(self, item):
        return int(item['class'].split('.')[0])

    def _get_bbox(self, image_path):
        return self.bbox[image_path[:-4]]


class FlowersDataset(BaseDataset):
    """datasets.py:608-642 (`img` key, integer class, no bounding boxes)."""


class PlacesSubSet(BaseDataset):
    """datasets.py:567-605: seven indoor scene classes named by the second path component of `image`
    ("./<class>/<file>"); `_get_img` drops the leading "./".  The reference's class reads attributes that its base
    class never sets (`image_embedding`, `prepare_training_pairs`) and cannot run; this follows BaseDataset's working
    train / test item layout instead."""
    class_label = {'bedroom': 0, 'dinette': 1, 'dining_room': 2, 'home_office': 3, 'hotel_room': 4,
                   'kitchenette': 5, 'living_room': 6}

    def __init__(self, data_root, train=True, base_size=64, transform=None, target_transform=None,
                 feature_switch="cnn_googlenet", device_normalize=False):
        super().__init__(data_root, train, base_size, transform, target_transform, feature_switch, device_normalize)

    def _get_img(self, item):
        return item['image'][2:]

    def _get_class(self, item):
        return self.class_label[item['image'].split("/")[1]]


def make_dataloader(dataset, batch_size, distributed=False, workers=0, shuffle=True, rank=None, world_size=None):
    """main.py:166-181: DistributedSampler shards the index set per rank; the default collate turns the per-branch
    image lists into per-branch batches."""
    param = {"num_workers": min(int(workers), os.cpu_count() or 1), "pin_memory": torch.cuda.is_available()}
    if distributed:
        kw = {}
        if rank is not None:
            kw = {"rank": rank, "num_replicas": world_size}
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, **kw)
        return data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, **param)
    return data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, **param)
