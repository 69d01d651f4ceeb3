"""Global configuration with the surface of the reference's StackGAN_v2/miscc/config.py:9-111.

`cfg` is one process-wide attribute dictionary; `cfg_from_file(path)` merges a YAML file into it,
rejecting unknown keys (KeyError) and type changes (ValueError) exactly as the reference's
`_merge_a_into_b` does (config.py:72-102).  Differences, on purpose: no easydict dependency, and the
YAML is read with the safe loader (the reference's bare `yaml.load(f)` fails under PyYAML >= 6).
"""
import numpy as np


class AttrDict(dict):
    """dict with attribute access; nested dicts become AttrDicts."""

    def __init__(self, mapping=None, **kw):
        super().__init__()
        for k, v in dict(mapping or {}, **kw).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, AttrDict):
            value = AttrDict(value)
        super().__setitem__(key, value)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    __setattr__ = __setitem__


# name kept for callers that do `from miscc.config import edict`-style things on the reference
edict = AttrDict


def _defaults():
    return AttrDict({
        "PROJECT_ROOT": ".", "BIRDS_DATA_ROOT": ".", "BIRDS_EMBEDDING_ROOT": ".",
        "DATASET_NAME": "birds", "EMBEDDING_TYPE": "cnn-rnn", "CONFIG_NAME": "", "DATA_DIR": "",
        "GPU_ID": "0", "CUDA": True, "WORKERS": 6,
        "TREE": {"BRANCH_NUM": 3, "BASE_SIZE": 64},
        "INCEPTION_CUB": False,
        "TEST": {"B_EXAMPLE": True, "SAMPLE_NUM": 30000, "TRUNC": 0.5},
        "TRAIN": {
            "BATCH_SIZE": 64, "VIS_COUNT": 64, "MAX_EPOCH": 600, "SNAPSHOT_INTERVAL": 2000,
            "DISCRIMINATOR_LR": 2e-4, "GENERATOR_LR": 2e-4, "FLAG": True, "NET_G": "", "NET_D": "",
            "COEFF": {"KL": 2.0, "CAL_LOSS": 0.0, "UNCOND_LOSS": 0.0, "COLOR_LOSS": 0.0},
            "LOG_INTERVAL": 10,
        },
        "GAN": {"EMBEDDING_DIM": 128, "DF_DIM": 64, "GF_DIM": 64, "Z_DIM": 100,
                "NETWORK_TYPE": "default", "R_NUM": 2, "B_CONDITION": True},
        "TEXT": {"DIMENSION": 1024},
    })


cfg = _defaults()
__C = cfg


def cfg_reset():
    """Restore the defaults in place (every module holds a reference to the same object)."""
    fresh = _defaults()
    cfg.clear()
    for k, v in fresh.items():
        cfg[k] = v
    return cfg


def _merge_a_into_b(a, b):
    if not isinstance(a, dict):
        return
    for k, v in a.items():
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        old = b[k]
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        if type(old) is not type(v):
            if isinstance(old, np.ndarray):
                v = np.array(v, dtype=old.dtype)
            else:
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(old), type(v), k))
        if isinstance(v, AttrDict):
            try:
                _merge_a_into_b(v, old)
            except Exception:
                print('Error under config key: {}'.format(k))
                raise
        else:
            b[k] = v


def cfg_from_file(filename):
    """Load a YAML config file and merge it into the defaults."""
    import yaml
    with open(filename, 'r') as f:
        loaded = yaml.safe_load(f)
    _merge_a_into_b(AttrDict(loaded or {}), cfg)


def cfg_from_dict(mapping):
    _merge_a_into_b(AttrDict(mapping), cfg)
