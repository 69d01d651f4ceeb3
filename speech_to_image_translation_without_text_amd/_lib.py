"""ctypes binding of the C-ABI in include/s2i_hip.h (libs2i_hip.so, built by csrc/Makefile).

There is no CPU fallback: if the shared library is missing, or a call is made on a tensor that
is not resident on a gfx950 device, this module raises.  torch must be imported first so that the
library's libamdhip64 dependency resolves to the HIP runtime torch already loaded.
"""
import ctypes
import os

import torch  # noqa: F401  (loads the HIP runtime the kernels attach to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libs2i_hip.so")

# enums of include/s2i_hip.h
CONV_K1, CONV_K3S1, CONV_K4S2, TCONV_K4S2, CONV_1D = 0, 1, 2, 3, 4
ACT_NONE, ACT_GLU, ACT_LRELU, ACT_TANH, ACT_RELU = 0, 1, 2, 3, 4
PACK_PLAIN, PACK_UPFOLD = 0, 1
DT_F32, DT_BF16 = 0, 1
ABI_VERSION = 4

c_int, c_float, c_void_p, c_size_t, c_ll = (ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                                            ctypes.c_size_t, ctypes.c_longlong)


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("kind", "B", "H", "W", "Cx", "Cc", "N", "wmode", "flip", "wR",
                                     "ldw", "act", "stats", "ldy", "groups", "nosplit", "kw", "stride", "pad",
                                     "tile_rows", "in_act", "in_groups")]


class WgradDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("kind", "B", "H", "W", "Ca", "Cc", "N", "ldg", "swap", "fold",
                                     "O", "I", "KH", "KW", "accumulate", "i_off", "I_total", "a_act", "a_groups")]


class PackItem(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("packed", c_void_p)] + [(n, c_int) for n in ("O", "I", "KH", "KW", "Ip", "mode", "gx",
                                                                               "block0")]


class Pack16Item(ctypes.Structure):
    _fields_ = [("P", c_void_p), ("out", c_void_p)] + [(n, c_int) for n in (
        "R", "C", "kind", "flip", "transpose", "T", "nphase", "Nn", "Npad", "Kk", "CK", "gx", "gy", "block0")]


P = c_void_p
_SIGNATURES = {
    "s2i_last_error": (ctypes.c_char_p, []),
    "s2i_version": (c_int, []),
    "s2i_check_device": (c_int, []),
    "s2i_set_tuning": (c_int, [ctypes.c_char_p, c_int]),
    "s2i_get_tuning": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_int)]),
    "s2i_conv_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "s2i_conv_stat_parts": (c_int, [ctypes.POINTER(ConvDesc)]),
    "s2i_conv_forward": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P, P, P, c_size_t, P]),
    "s2i_conv_forward_cls": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P, P, P, P, c_size_t, P]),
    "s2i_conv_forward_in": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P, P, c_size_t, P]),
    "s2i_conv_wgrad_in_eligible": (c_int, [ctypes.POINTER(WgradDesc)]),
    "s2i_conv_wgrad_in": (c_int, [ctypes.POINTER(WgradDesc), P, P, P, P, P, c_size_t, P]),
    "s2i_conv_split_eligible": (c_int, [ctypes.POINTER(ConvDesc)]),
    "s2i_conv_forward_split": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, c_int, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "s2i_split_packed_weight": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P]),
    "s2i_wgrad_workspace_bytes_split": (c_size_t, [ctypes.POINTER(WgradDesc), c_int]),
    "s2i_conv_wgrad_split": (c_int, [ctypes.POINTER(WgradDesc), c_int, P, P, P, P, P, c_size_t, P]),
    "s2i_conv_bf16_eligible": (c_int, [ctypes.POINTER(ConvDesc)]),
    "s2i_conv_bf16_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "s2i_conv_bf16_stat_parts": (c_int, [ctypes.POINTER(ConvDesc)]),
    "s2i_conv_bf16_weight_elems": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "s2i_conv_bf16_weight_layout": (c_int, [ctypes.POINTER(ConvDesc)]),
    "s2i_pack16_item_fill": (c_int, [ctypes.POINTER(ConvDesc), P, c_int, c_int, P, ctypes.POINTER(Pack16Item)]),
    "s2i_pack_conv_weights_bf16_batched": (c_int, [P, c_int, c_int, P]),
    "s2i_pack_conv_weight_bf16": (c_int, [ctypes.POINTER(ConvDesc), P, c_int, c_int, P, P]),
    "s2i_conv_forward_bf16": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P, P, c_size_t, P]),
    "s2i_conv_forward_dt": (c_int, [ctypes.POINTER(ConvDesc), P, c_int, P, P, P, P, P, c_int, P, P, c_size_t, P]),
    "s2i_wgrad_workspace_bytes_dt": (c_size_t, [ctypes.POINTER(WgradDesc), c_int, c_int]),
    "s2i_conv_wgrad_dt": (c_int, [ctypes.POINTER(WgradDesc), P, c_int, P, P, c_int, P, P, c_size_t, P]),
    "s2i_bn_act_forward_dt": (c_int, [c_int, P, c_ll, c_int, c_int, P, c_int, P, P, P]),
    "s2i_bn_act_bwd_reduce_dt": (c_int, [c_int, P, P, c_int, c_ll, c_int, c_int, P, c_int, P, c_int, P]),
    "s2i_bn_act_bwd_apply_dt": (c_int, [c_int, P, P, c_int, c_ll, c_int, c_int, P, P, c_int, P, P]),
    "s2i_act_backward_dt": (c_int, [c_int, P, P, c_int, c_ll, c_int, c_int, P, P]),
    "s2i_nchw_to_nhwc_dt": (c_int, [c_int, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "s2i_nhwc_to_nchw_dt": (c_int, [c_int, P, c_int, P, c_int, c_int, c_int, c_int, P]),
    "s2i_spatial_sum_dt": (c_int, [c_int, P, c_int, c_int, c_int, c_int, P, P, c_size_t, P]),
    "s2i_tap_sums_dt": (c_int, [c_int, P, c_int, c_int, c_int, c_int, P, P, c_size_t, P]),
    "s2i_cast": (c_int, [P, c_int, P, c_int, c_ll, P]),
    "s2i_cvec_bias_table": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_size_t, P]),
    "s2i_border_sums_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "s2i_tap_sums": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_size_t, P]),
    "s2i_cvec_grads": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P]),
    "s2i_wgrad_workspace_bytes": (c_size_t, [ctypes.POINTER(WgradDesc)]),
    "s2i_conv_wgrad": (c_int, [ctypes.POINTER(WgradDesc), P, P, P, P, P, c_size_t, P]),
    "s2i_pack_conv_weight": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "s2i_pack_conv_weights_batched": (c_int, [P, c_int, c_int, c_int, P]),
    "s2i_bn_finalize": (c_int, [P, c_int, c_int, c_int, c_ll, P, P, P, P, P, c_float, c_float, P, P]),
    "s2i_bn_eval_coeffs": (c_int, [c_int, P, P, P, P, c_float, P, P]),
    "s2i_bn_act_forward": (c_int, [P, c_ll, c_int, c_int, P, c_int, P, P, P]),
    "s2i_colstats": (c_int, [P, c_ll, c_int, c_int, P, c_int, P]),
    "s2i_bn_act_bwd_reduce": (c_int, [P, P, c_int, c_ll, c_int, c_int, P, c_int, P, c_int, P]),
    "s2i_bn_bwd_finalize": (c_int, [P, c_int, c_int, c_int, c_ll, P, P, c_int, P, P]),
    "s2i_bn_act_bwd_apply": (c_int, [P, P, c_int, c_ll, c_int, c_int, P, P, c_int, P, P]),
    "s2i_act_backward": (c_int, [P, P, c_int, c_ll, c_int, c_int, P, P]),
    "s2i_glu_forward": (c_int, [P, c_ll, c_int, P, P]),
    "s2i_glu_backward": (c_int, [P, P, c_ll, c_int, P, P]),
    "s2i_nchw_to_nhwc": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "s2i_nhwc_to_nchw": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, P]),
    "s2i_image_to_u8": (c_int, [P, c_int, P, c_ll, P]),
    "s2i_u8_to_image": (c_int, [P, P, c_int, c_int, c_int, P]),
    "s2i_spatial_sum": (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_size_t, P]),
    "s2i_spatial_sum_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "s2i_reparam_forward": (c_int, [P, P, c_int, c_int, P, P]),
    "s2i_reparam_backward": (c_int, [P, P, P, P, P, c_int, c_int, P, P]),
    "s2i_kl_forward": (c_int, [P, c_int, P, c_int, c_int, c_int, P, P]),
    "s2i_kl_backward": (c_int, [P, c_int, P, c_int, c_int, c_int, P, P, P, P]),
    "s2i_logit_forward": (c_int, [P, P, P, c_int, c_int, P, P]),
    "s2i_logit_backward": (c_int, [P, P, P, P, c_int, c_int, P, c_int, P, P, c_int, P]),
    "s2i_bce_forward": (c_int, [P, c_float, c_int, c_float, P, c_int, P]),
    "s2i_bce_backward": (c_int, [P, c_float, c_int, c_float, P, P, P]),
    "s2i_bce_multi_forward": (c_int, [P, P, P, c_int, c_int, c_int, P, P]),
    "s2i_bce_multi_backward": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P]),
    "s2i_cal_loss": (c_int, [P, P, c_int, c_int, P, c_int, P, P]),
    "s2i_maxpool_w3s2": (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    "s2i_lstm_cell": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P]),
    "s2i_lstm_step": (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_int, P]),
    "s2i_time_mean": (c_int, [P, c_int, c_int, c_int, P, P]),
    "s2i_adam_step": (c_int, [P, P, P, P, c_ll, c_float, c_float, c_float, c_float, c_int, P, c_float, P]),
    "s2i_increment": (c_int, [P, P]),
    "s2i_ema_update": (c_int, [P, P, c_ll, c_float, P]),
    "s2i_scale_dev": (c_int, [P, P, c_ll, P, P]),
    "s2i_axpby": (c_int, [P, P, c_ll, c_float, c_float, P]),
    "s2i_plan_create": (c_int, [P, ctypes.POINTER(P), ctypes.POINTER(c_int)]),
    "s2i_plan_replay": (c_int, [P, P]),
    "s2i_plan_destroy": (c_int, [P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None
_device_checked = False


class S2IError(RuntimeError):
    pass


def load():
    """Load libs2i_hip.so (once) and declare every entry point of include/s2i_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise S2IError(
            "libs2i_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C %s/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback."
            % (LIB_PATH, _HERE))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.s2i_version() != ABI_VERSION:
        raise S2IError("libs2i_hip.so ABI version %d, expected %d: rebuild it (make -C csrc)" % (lib.s2i_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().s2i_last_error()
        raise S2IError("%s failed: %s" % (what, msg.decode() if msg else "unknown error"))


class tuning:
    """`with tuning(fwd_bm=96): ...` sets integer knobs of the launch planners (s2i_set_tuning) for the body and restores
    them: tools and tests use it to force a kernel variant; nothing on a launch path reads the environment."""

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        lib = load()
        self.old = {}
        for k, v in self.knobs.items():
            cur = c_int(0)
            check(lib.s2i_get_tuning(k.encode(), ctypes.byref(cur)), "s2i_get_tuning")
            self.old[k] = cur.value
            check(lib.s2i_set_tuning(k.encode(), int(v)), "s2i_set_tuning")
        return self

    def __exit__(self, *exc):
        lib = load()
        for k, v in self.old.items():
            check(lib.s2i_set_tuning(k.encode(), v), "s2i_set_tuning")
        return False


def require_device():
    """Raise unless the current device is a gfx950 GPU the kernels were compiled for."""
    global _device_checked
    if _device_checked:
        return
    if not torch.cuda.is_available():
        raise S2IError("no HIP device visible: the MI355X kernels have no CPU fallback")
    check(load().s2i_check_device(), "s2i_check_device")
    _device_checked = True


def ptr(t):
    """Device pointer of a contiguous fp32/int32 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise S2IError("tensor is not on a HIP device; the MI355X path has no CPU fallback")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_dev = torch.cuda.current_device


def stream():
    """Raw hipStream_t of torch's current stream.  `torch.cuda.current_stream().cuda_stream` builds a Stream object per
    call (8 us; 1 600 calls per train step = 2.6 ms of host time, tools/host_profile.py); the raw query is ~0.3 us."""
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())
    return torch.cuda.current_stream().cuda_stream
