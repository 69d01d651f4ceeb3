"""CPU suite for the host side: the C-ABI library loads and exports every symbol include/s2i_hip.h declares,
the config surface behaves like the reference's (miscc/config.py:72-111), module / checkpoint layout, flat
parameter storage.  No compute calls (no GPU here)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from speech_to_image_translation_without_text_amd import _lib
    header = open(os.path.join(ROOT, "include", "s2i_hip.h")).read()
    declared = set(re.findall(r"\b(s2i_[a-z0-9_]+)\s*\(", header))
    declared -= {"s2i_conv_desc", "s2i_wgrad_desc"}
    lib = _lib.load()
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, "declared in s2i_hip.h but not exported: %s" % missing
    unbound = sorted(declared - set(_lib.EXPORTED_SYMBOLS))
    assert not unbound, "declared in s2i_hip.h but not bound in _lib.py: %s" % unbound
    assert lib.s2i_version() == _lib.ABI_VERSION == 4


def test_descriptor_validation_reports_errors_without_a_gpu():
    """Planning functions are host code: bad shapes come back as an error string, not a crash."""
    import ctypes
    from speech_to_image_translation_without_text_amd import _lib
    lib = _lib.load()
    bad = _lib.ConvDesc(_lib.CONV_K3S1, 2, 6, 6, 8, 0, 16, 0, 0, 8, 16, 0, 0, 16, 1, 0)  # 6x6 is not a power of two
    assert lib.s2i_conv_stat_parts(ctypes.byref(bad)) == -1
    assert b"powers of two" in lib.s2i_last_error()
    odd = _lib.ConvDesc(_lib.CONV_K3S1, 2, 8, 8, 6, 0, 16, 0, 0, 6, 16, 0, 0, 16, 1, 0)  # 6 channels
    assert lib.s2i_conv_stat_parts(ctypes.byref(odd)) == -1
    assert b"multiples of 4" in lib.s2i_last_error()
    ok = _lib.ConvDesc(_lib.CONV_K3S1, 2, 8, 8, 8, 0, 16, 0, 0, 8, 16, 0, 1, 16, 1, 0)
    assert lib.s2i_conv_stat_parts(ctypes.byref(ok)) == 1
    # split-bf16 entry points: eligibility is a host-side rule (gathered channels in whole 32-deep chunks, not the RGB path)
    thin = _lib.ConvDesc(_lib.CONV_K3S1, 2, 8, 8, 8, 0, 16, 0, 0, 8, 16, 0, 0, 16, 1, 0)
    assert lib.s2i_conv_split_eligible(ctypes.byref(thin)) == 0
    wide = _lib.ConvDesc(_lib.CONV_K3S1, 2, 8, 8, 64, 0, 128, 0, 0, 64, 128, 0, 1, 128, 1, 0)
    assert lib.s2i_conv_split_eligible(ctypes.byref(wide)) == 1
    assert lib.s2i_conv_split_eligible(ctypes.byref(bad)) == 0
    # weight-gradient planning: same workspace rule in both modes, errors as strings
    wd = _lib.WgradDesc(_lib.CONV_K3S1, 2, 8, 8, 64, 0, 128, 128, 0, 0, 128, 64, 3, 3, 0, 0, 0)
    assert lib.s2i_wgrad_workspace_bytes(ctypes.byref(wd)) > 0
    assert lib.s2i_wgrad_workspace_bytes_split(ctypes.byref(wd), 3) > 0
    wbad = _lib.WgradDesc(_lib.CONV_K3S1, 2, 6, 6, 64, 0, 128, 128, 0, 0, 128, 64, 3, 3, 0, 0, 0)
    assert lib.s2i_wgrad_workspace_bytes_split(ctypes.byref(wbad), 3) == 0
    assert b"powers of two" in lib.s2i_last_error()
    # compute entry points refuse bad arguments before touching a device
    assert lib.s2i_conv_forward_split(ctypes.byref(wide), None, None, None, 3, 128, 64, None, None, None, None, None, 0,
                                      None) != 0
    assert b"planes" in lib.s2i_last_error()
    assert lib.s2i_lstm_step(None, 0, None, None, None, 4, 8, 512, 2, 0, None, None, None, None, 0, None) != 0


def test_no_cpu_fallback():
    from speech_to_image_translation_without_text_amd import _lib, model
    from helpers import CASES, configure
    configure(CASES['small3'])
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = model.G_NET()
    with pytest.raises(_lib.S2IError):
        g(torch.randn(2, 12), torch.randn(2, 32))


def test_config_merge_rules(tmp_path):
    from speech_to_image_translation_without_text_amd.miscc.config import cfg, cfg_from_file, cfg_reset
    cfg_reset()
    good = tmp_path / "good.yml"
    good.write_text("TREE:\n  BRANCH_NUM: 2\nTRAIN:\n  BATCH_SIZE: 8\n  COEFF:\n    UNCOND_LOSS: 1.0\n")
    cfg_from_file(str(good))
    assert cfg.TREE.BRANCH_NUM == 2 and cfg.TRAIN.BATCH_SIZE == 8 and cfg.TRAIN.COEFF.UNCOND_LOSS == 1.0
    unknown = tmp_path / "unknown.yml"
    unknown.write_text("NOT_A_KEY: 1\n")
    with pytest.raises(KeyError):
        cfg_from_file(str(unknown))
    wrong_type = tmp_path / "type.yml"
    wrong_type.write_text("TRAIN:\n  BATCH_SIZE: 'eight'\n")
    with pytest.raises(ValueError):
        cfg_from_file(str(wrong_type))
    cfg_reset()
    cfg_from_file(os.path.join(ROOT, "speech_to_image_translation_without_text_amd", "cfg", "birds_3stages.yml"))
    assert cfg.TRAIN.BATCH_SIZE == 24 and cfg.GAN.R_NUM == 2 and cfg.TRAIN.COEFF.CAL_LOSS == 50.0
    cfg_reset()


def test_wrapper_state_dict_has_module_prefix_and_flat_storage():
    from helpers import CASES, build_nets
    from speech_to_image_translation_without_text_amd import trainer as T
    netG, netsD = build_nets(CASES['small3'])
    wrapped = T._Replica(netsD[0], [0])
    keys = list(wrapped.state_dict().keys())
    assert keys[0] == 'module.img_code_s16.0.weight' and all(k.startswith('module.') for k in keys)
    before = {k: v.clone() for k, v in netG.state_dict().items()}
    flat = T.FlatNet(netG, 2e-4, with_ema=True)
    for k, v in netG.state_dict().items():
        assert torch.equal(v, before[k]), k          # re-homing keeps every value
    for p, o, n in zip(flat.params, flat.offsets, flat.sizes):
        assert o % 4 == 0 and p.data_ptr() == flat.p.data_ptr() + 4 * o and p.grad.data_ptr() == flat.g.data_ptr() + 4 * o
    assert sum(flat.sizes) == sum(p.numel() for p in netG.parameters())
    assert torch.equal(flat.avg, flat.p)


def test_weights_init_matches_reference_rules():
    from speech_to_image_translation_without_text_amd import trainer as T
    conv = torch.nn.Conv2d(8, 16, 3)
    bn = torch.nn.BatchNorm2d(16)
    lin = torch.nn.Linear(12, 6)
    torch.manual_seed(0)
    for m in (conv, bn, lin):
        T.weights_init(m)
    w = conv.weight.view(16, -1)
    assert torch.allclose(w @ w.t(), torch.eye(16), atol=1e-5)          # orthogonal rows
    assert float(bn.bias.abs().max()) == 0.0 and abs(float(bn.weight.mean()) - 1.0) < 0.05
    assert float(lin.bias.abs().max()) == 0.0


def test_header_is_valid_c(tmp_path):
    """include/s2i_hip.h is the drop-in boundary: it must compile as plain C (no C++-only constructs, every declaration at
    file scope)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "h.c"
    src.write_text('#include "s2i_hip.h"\nint main(void) { s2i_conv_desc c; s2i_wgrad_desc w; (void)c; (void)w; return 0; }\n')
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-c", str(src), "-o",
                    str(tmp_path / "h.o")], check=True)


def test_tuning_knobs_roundtrip_without_a_gpu():
    """s2i_set_tuning / s2i_get_tuning (the planners' integer knobs; the library reads no environment variable on a launch
    path) and the descriptor's tile_rows: unknown keys and values are rejected with a message."""
    import ctypes
    from speech_to_image_translation_without_text_amd import _lib
    lib = _lib.load()
    with _lib.tuning(fwd_bm=96, b16_v2=2):
        v = ctypes.c_int(0)
        assert lib.s2i_get_tuning(b"fwd_bm", ctypes.byref(v)) == 0 and v.value == 96
    v = ctypes.c_int(0)
    assert lib.s2i_get_tuning(b"fwd_bm", ctypes.byref(v)) == 0 and v.value == -1     # restored: unset
    assert lib.s2i_set_tuning(b"no_such_knob", 1) != 0 and b"no_such_knob" in lib.s2i_last_error()
    d = _lib.ConvDesc(_lib.CONV_K3S1, 2, 8, 8, 64, 0, 128, 0, 0, 64, 128, 0, 1, 128, 1, 1, 0, 0, 0, 64)
    assert lib.s2i_conv_stat_parts(ctypes.byref(d)) < 0 and b"tile_rows" in lib.s2i_last_error()
    d.tile_rows = 96          # 128 rows: too few for two 96-row tiles, the planner keeps 128
    assert lib.s2i_conv_stat_parts(ctypes.byref(d)) == 1
    big = _lib.ConvDesc(_lib.CONV_K4S2, 72, 64, 64, 128, 0, 256, 0, 0, 128, 256, 0, 1, 256, 3, 0, 0, 0, 0, 0)
    assert lib.s2i_conv_stat_parts(ctypes.byref(big)) == 72 * 32 * 32 // 96      # the stacked passes take 96-row tiles
    big.tile_rows = 128
    assert lib.s2i_conv_stat_parts(ctypes.byref(big)) == 72 * 32 * 32 // 128


def test_weight_gradient_planner_tile_heights_without_a_gpu():
    """The weight-gradient planner's three bf16 tile shapes and two fp32 ones (s2i_igemm.hip::plan_wgrad) through the
    workspace query: every plan asks for a whole number of K x N fp32 slabs, a forced shape is honoured only where 256
    divides the taps x channels rows (and, for 256 x 256, the output channels), and the cost model picks the large tiles on
    D_NET256's stacked stride-2 layers."""
    import ctypes
    from speech_to_image_translation_without_text_amd import _lib
    lib = _lib.load()
    BF16 = 1

    def slabs(d, dt, **knobs):
        with _lib.tuning(**knobs):
            n = lib.s2i_wgrad_workspace_bytes_dt(ctypes.byref(d), dt, dt)
        kn = (16 if d.kind == _lib.CONV_K4S2 else 9) * d.Ca * d.N * 4
        assert n > 0 and n % kn == 0, (n, kn)
        return n // kn

    # D_NET256 conv3 over the stacked batch of config 4: a (144,64,64,128), g (144,32,32,256)
    d = _lib.WgradDesc(_lib.CONV_K4S2, 144, 64, 64, 128, 0, 256, 256, 0, 0, 256, 128, 4, 4, 1, 0, 0, 0, 0)
    s128, s256, s512, s0 = (slabs(d, BF16, wgrad16_bm=v) for v in (128, 256, 512, 0))
    # K = 2048 rows x 256 columns: 32 / 16 / 8 tiles; tiles x splits fill the chip's 768 / 512 / 256 block slots
    assert 700 <= s128 * 32 <= 1600 and 400 <= s256 * 16 <= 1100 and 200 <= s512 * 8 <= 600, (s128, s256, s512)
    assert s0 in (s256, s512)                                                     # the model leaves 128 x 128 here
    f = {v: slabs(d, 0, wgrad_bm=v) for v in (128, 256, 512, 0)}
    assert 700 <= f[128] * 32 <= 1600 and 400 <= f[256] * 16 <= 1100 and 200 <= f[512] * 8 <= 600 and f[0] in (f[256], f[512]), f
    # 3x3, 64 -> 128 channels: 576 rows, 256 does not divide them: every setting plans the same 128-row tiles
    e = _lib.WgradDesc(_lib.CONV_K3S1, 48, 64, 64, 64, 0, 128, 128, 0, 0, 128, 64, 3, 3, 1, 0, 0, 0, 0)
    assert len({slabs(e, BF16, wgrad16_bm=v) for v in (0, 128, 256, 512)}) == 1
