"""Host-side operators of the MI355X StackGAN-v2 path: torch.autograd.Functions whose forward and
backward are calls into the C-ABI kernels of include/s2i_hip.h (through _lib.py).

Each Function stands in for a group of stock torch ops of the reference
(StackGAN_v2/model.py:112-551, StackGAN_v2/trainer.py:54-58, 298-311, 394-409):

  ConvBnAct   conv (1x1 / 3x3 / 4x4-s2 / nearest-x2+3x3) + BatchNorm(train) + GLU|LeakyReLU|none
              (+ residual add): upBlock, Block3x3_relu, ResBlock halves, downBlock,
              Block3x3_leakRelu, INIT_STAGE_G.fc
  ConvAct     conv + bias + LeakyReLU|tanh|none: first D conv, GET_IMAGE_G, CA_NET.fc
  Glu2d, Reparam, KLLoss        CA_NET encode / reparametrize, KL_loss
  LogitHead, BCELoss, ClassAwareLoss   logits / uncond_logits + nn.BCELoss, class_aware_loss
  ToNHWC / ToNCHW               layout changes at the module boundary

Activations between these ops are NHWC; parameters stay in the reference's OIHW layout and are
re-packed to the kernels' layout when their version counter changes.  There is no CPU fallback.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (ACT_GLU, ACT_LRELU, ACT_NONE, CONV_K1, CONV_K3S1, CONV_K4S2, DT_BF16, DT_F32,
                   PACK_PLAIN, PACK_UPFOLD, TCONV_K4S2, ConvDesc, WgradDesc, check, ptr, stream)

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# When True (set by the fused trainer), parameter gradients are accumulated by the kernels straight into
# `param.grad` (the flat gradient buffer's views) and backward returns None for them: no temporary
# gradient tensors, no autograd AccumulateGrad add kernels.  Requires zeroed, pre-attached `.grad`.
DIRECT_PARAM_GRAD = False


def _direct(p):
    return (DIRECT_PARAM_GRAD and p is not None and p.requires_grad and p.grad is not None
            and p.grad.is_contiguous())


class param_grad_mode:
    """Scope of the module-level switch: `with param_grad_mode(direct=True): ...` turns the direct accumulation into
    `.grad` on for the body and restores the previous value afterwards, so a later stock-optimiser use of the same
    modules in this process sees the default (autograd-returned gradients)."""

    def __init__(self, direct=True):
        self.new = bool(direct)

    def __enter__(self):
        global DIRECT_PARAM_GRAD
        self.old = DIRECT_PARAM_GRAD
        DIRECT_PARAM_GRAD = self.new
        return self

    def __exit__(self, *exc):
        global DIRECT_PARAM_GRAD
        DIRECT_PARAM_GRAD = self.old
        return False


def _lib_ready():
    lib = _lib.load()
    _lib.require_device()
    return lib


# ---- scratch ------------------------------------------------------------------------------------
class _Workspace:
    """One growable scratch buffer per (device, stream): kernels on a stream use it one after another, and two
    streams never share one.  Once a hipGraph has been captured (`pin_graph_resources`) a buffer that has to grow is
    replaced but never freed: the graph keeps writing split-K slabs through the raw pointer it captured."""

    def __init__(self):
        self.buf = {}
        self.pinned = False
        self.retired = []

    def get(self, nbytes, device):
        nbytes = max(int(nbytes), 16)
        key = (device, stream())     # the raw query: a Stream object per scratch request cost 8 us (tools/host_profile.py)
        cur = self.buf.get(key)
        if cur is None or cur.numel() * 4 < nbytes:
            if cur is not None and self.pinned:
                self.retired.append(cur)
            # during hipGraph capture the allocation comes from the graph's private pool and stays reserved for it
            cur = torch.empty((nbytes + 3) // 4 + 1024, dtype=torch.float32, device=device)
            self.buf[key] = cur
        return cur


_ws = _Workspace()


def pin_graph_resources():
    """Called before a hipGraph capture: everything a captured launch reaches through a raw pointer that lives outside
    the graph's own memory pool -- the per-stream workspaces, the device tables of the batched weight packs -- is kept
    alive for the rest of the process (replaced workspaces are retired, not freed; the table caches are no longer cleared)."""
    _ws.pinned = True


def invalidate_derived(params):
    """Drop every weight copy derived from the packed fp32 tensors of these parameters (bf16 stage layouts, split bf16
    planes): they are re-derived from the current packed weights on first use.  Needed before an eager step that follows
    hipGraph replays, which update the masters and re-pack on the device without running the host-side bookkeeping."""
    for p in params:
        packs = getattr(p, '_s2i_packs', None)
        if packs:
            for ent in packs.values():
                ent[0]._s2i_gen = getattr(ent[0], '_s2i_gen', 0) + 1


def _roundup4(v):
    return (v + 3) & ~3


# ---- packed weights ------------------------------------------------------------------------------
# OIHW parameter -> P[tap][Ip][Op].  The packed copy lives ON the parameter object (attribute
# `_s2i_packs`), so it dies with the parameter and can never be mistaken for another network's; it is
# refreshed when torch's version counter or the storage address changes, or explicitly after an update
# that bypasses the version counter (the fused Adam kernel).
def packed_weight(w, mode=0):
    packs = getattr(w, '_s2i_packs', None)
    if packs is None:
        packs = {}
        w._s2i_packs = packs
    ent = packs.get(mode)
    ver, addr = w._version, w.data_ptr()
    if ent is not None and ent[1] == ver and ent[2] == addr and ent[0].device == w.device:
        return ent[0]
    out = ent[0] if ent is not None and ent[0].device == w.device else None
    packed = pack_weight(w, mode, out=out)
    packs[mode] = (packed, ver, addr)
    return packed


_pack_tables = {}


def refresh_packed(params):
    """Re-pack every cached layout of these parameters in place -- one launch for the whole list (the table of
    (parameter, packed copy, shape) entries lives on the device and is rebuilt only when a pointer changes)."""
    entries = []
    for p in params:
        packs = getattr(p, '_s2i_packs', None)
        if packs:
            for mode, ent in list(packs.items()):
                if ent[0].device == p.device:
                    entries.append((p, mode, ent[0]))
                else:
                    del packs[mode]
    if not entries:
        return
    lib = _lib_ready()
    key = tuple((p.data_ptr(), mode, out.data_ptr()) for p, mode, out in entries)
    tab = _pack_tables.get(key)
    if tab is None:
        items = (_lib.PackItem * len(entries))()
        block0, max_taps = 0, 1
        for k, (p, mode, out) in enumerate(entries):
            if p.dim() == 2:
                O, I, KH, KW = p.shape[0], p.shape[1], 1, 1
            else:
                O, I, KH, KW = p.shape
            Ip, Op = _roundup4(I), _roundup4(O)
            gx = (Op + 63) // 64          # 64-channel tiles (csrc/s2i_igemm.hip::pack_tile)
            items[k] = _lib.PackItem(p.data_ptr(), out.data_ptr(), O, I, KH, KW, Ip, mode, gx, block0)
            block0 += gx * ((Ip + 7) // 8)
            max_taps = max(max_taps, KH * KW)
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(entries[0][0].device)
        tab = (key, raw, len(entries), block0, max_taps)
        if len(_pack_tables) > 64 and not _ws.pinned:   # a captured graph reads its table through a raw pointer
            _pack_tables.clear()
        _pack_tables[key] = tab
    check(lib.s2i_pack_conv_weights_batched(ptr(tab[1]), tab[2], tab[3], tab[4], stream()), "s2i_pack_conv_weights_batched")
    for p, mode, out in entries:
        out._s2i_gen = getattr(out, '_s2i_gen', 0) + 1
        p._s2i_packs[mode] = (out, p._version, p.data_ptr())
    _refresh_bf16([out for _, _, out in entries])


_pack16_tables = {}


def _refresh_bf16(packed_list):
    """Every bf16 copy that was ever derived from these packed fp32 tensors (bf16_weight below) re-derived in ONE launch,
    instead of one 10 us launch per layer on its first use after the optimiser step (70 launches per step at config 4)."""
    recs = []
    for packed in packed_list:
        descs = getattr(packed, '_s2i_b16_desc', None)
        if descs:
            for key, (d, w_offset, ent) in descs.items():
                recs.append((packed, key, d, w_offset, ent))
    if not recs:
        return
    lib = _lib_ready()
    tkey = tuple((packed.data_ptr(), key, ent.data_ptr()) for packed, key, _, _, ent in recs)
    tab = _pack16_tables.get(tkey)
    if tab is None:
        items = (_lib.Pack16Item * len(recs))()
        block0 = 0
        for k, (packed, key, d, w_offset, ent) in enumerate(recs):
            nb = lib.s2i_pack16_item_fill(ctypes.byref(d), ptr(packed) + 4 * int(w_offset), packed.shape[1], packed.shape[2],
                                          ptr(ent), ctypes.byref(items[k]))
            if nb <= 0:
                check(1, "s2i_pack16_item_fill")
            items[k].block0 = block0
            block0 += nb
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(recs[0][0].device)
        tab = (raw, len(recs), block0)
        if len(_pack16_tables) > 64 and not _ws.pinned:
            _pack16_tables.clear()
        _pack16_tables[tkey] = tab
    check(lib.s2i_pack_conv_weights_bf16_batched(ptr(tab[0]), tab[1], tab[2], stream()), "s2i_pack_conv_weights_bf16_batched")
    for packed, key, d, w_offset, ent in recs:
        gen = getattr(packed, '_s2i_gen', 0)
        cache = getattr(packed, '_s2i_b16', None)
        if cache is None or cache[0] != gen:
            cache = (gen, {})
            packed._s2i_b16 = cache
        cache[1][key] = ent


def pack_weight(w, mode, out=None):
    lib = _lib_ready()
    w = w.detach()
    if w.dim() == 2:
        O, I, KH, KW = w.shape[0], w.shape[1], 1, 1
    else:
        O, I, KH, KW = w.shape
    if not w.is_contiguous():
        w = w.contiguous()
    T = 16 if mode == PACK_UPFOLD else KH * KW
    Ip, Op = _roundup4(I), _roundup4(O)
    if out is None or out.numel() != T * Ip * Op:
        out = torch.empty((T, Ip, Op), dtype=torch.float32, device=w.device)
    check(lib.s2i_pack_conv_weight(ptr(w), ptr(out), O, I, KH, KW, Ip, mode, stream()), "s2i_pack_conv_weight")
    out._s2i_gen = getattr(out, '_s2i_gen', 0) + 1  # invalidates the split-bf16 copies derived from it
    return out


# Matrix-product mode of the convolution GEMMs: 0 = native fp32 MFMA (default); 2 / 3 = operands split into that many
# bf16 planes (include/s2i_hip.h, "split-bf16 matrix products").  Opt-in: S2I_MATH_PLANES=2|3.
MATH_PLANES = int(os.environ.get("S2I_MATH_PLANES", "0"))


def split_weight(packed, planes, transpose):
    """bf16 planes of a packed weight tensor P[T][R][C], cached on the tensor object until it is re-packed.  Both operand
    layouts ([plane][T][R][C] for the input gradient, [plane][T][C][R] for the forward) come from one launch."""
    lib = _lib_ready()
    cache = getattr(packed, '_s2i_split', None)
    gen = packed._s2i_gen
    if cache is None or cache[0] != gen or cache[1] != planes:
        T, R, C = packed.shape
        rc, cr = (cache[2], cache[3]) if cache is not None and cache[1] == planes else (
            torch.empty((planes, T, R, C), dtype=torch.int16, device=packed.device),
            torch.empty((planes, T, C, R), dtype=torch.int16, device=packed.device))
        check(lib.s2i_split_packed_weight(ptr(packed), T, R, C, planes, ptr(rc), ptr(cr), stream()), "s2i_split_packed_weight")
        cache = (gen, planes, rc, cr)
        packed._s2i_split = cache
    return cache[3] if transpose else cache[2]


# ---- bf16 activation mode (BASELINE config 4) ------------------------------------------------------------------------
# ACT_BF16 = True (S2I_ACT_BF16=1, bench.py --math bf16): activations between the fused blocks are stored as bf16 NHWC,
# convolutions whose channel counts allow it run on the bf16 matrix cores from bf16 weights (s2i_conv_forward_bf16),
# BatchNorm statistics come from the fp32 accumulators, master weights / gradients / Adam / EMA stay fp32.  fp32 islands:
# CA_NET, INIT_STAGE_G's fc (+ BatchNorm1d), the NHWC4 image tensors, the logit heads and the losses.
ACT_BF16 = os.environ.get("S2I_ACT_BF16", "0") == "1"


def _dt(t):
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype != torch.float32:
        raise _lib.S2IError("activation tensors are fp32 or bf16, got %s" % t.dtype)
    return DT_F32


def cast(t, dtype):
    """Contiguous tensor in the other activation dtype (own kernel: torch is not on the product path for arithmetic)."""
    if t.dtype == dtype:
        return t
    lib = _lib_ready()
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=dtype, device=t.device)
    n = t.numel()
    if n % 4 != 0:
        raise _lib.S2IError("cast: %d elements (not a multiple of 4)" % n)
    check(lib.s2i_cast(ptr(t), _dt(t), ptr(out), _dt(out), n, stream()), "s2i_cast")
    return out


def bf16_weight(packed, d, w_offset):
    """bf16 weights of one convolution descriptor in the stage order of s2i_conv_forward_bf16, cached on the packed fp32
    tensor until it is re-packed (the fused Adam bumps `_s2i_gen`)."""
    lib = _lib_ready()
    cache = getattr(packed, '_s2i_b16', None)
    gen = getattr(packed, '_s2i_gen', 0)
    if cache is None or cache[0] != gen:
        cache = (gen, {})
        packed._s2i_b16 = cache
    key = (d.kind, d.wmode, d.flip, d.Cx, d.N, int(w_offset), lib.s2i_conv_bf16_weight_layout(ctypes.byref(d)))
    ent = cache[1].get(key)
    if ent is None:
        old = getattr(packed, '_s2i_b16_bufs', None)
        if old is None:
            old = packed._s2i_b16_bufs = {}
        ent = old.get(key)               # reuse the allocation across generations
        n = lib.s2i_conv_bf16_weight_elems(ctypes.byref(d))
        if n == 0:
            check(1, "s2i_conv_bf16_weight_elems")
        if ent is None or ent.numel() != n:
            ent = torch.empty(n, dtype=torch.bfloat16, device=packed.device)
            old[key] = ent
        check(lib.s2i_pack_conv_weight_bf16(ctypes.byref(d), ptr(packed) + 4 * int(w_offset), packed.shape[1],
                                            packed.shape[2], ptr(ent), stream()), "s2i_pack_conv_weight_bf16")
        cache[1][key] = ent
        # remembered for the batched re-derivation after the next optimiser step (_refresh_bf16)
        descs = getattr(packed, '_s2i_b16_desc', None)
        if descs is None:
            descs = packed._s2i_b16_desc = {}
        dd = ConvDesc()
        ctypes.memmove(ctypes.byref(dd), ctypes.byref(d), ctypes.sizeof(ConvDesc))
        descs[key] = (dd, int(w_offset), ent)
    return ent


def conv_any(kind, x, packed, N, *, wmode=0, flip=0, bias=None, act=ACT_NONE, stats=False, groups=1, w_offset=0,
             cls_bias=None, out_dtype=torch.float32):
    """Convolution of an NHWC tensor of either activation dtype (no broadcast vector) -> (y, part, nparts).  bf16 in /
    bf16 out goes to the patch-staged bf16 kernel when the layer is eligible, everything else to the fp32-MFMA kernel
    reading / writing the given dtypes."""
    lib = _lib_ready()
    B, H, W, Cx = x.shape
    Ho, Wo = _geom(kind, H, W)
    wR, ldw = packed.shape[1], packed.shape[2]
    d = ConvDesc(kind, B, H, W, Cx, 0, N, wmode, flip, wR, ldw, act, 1 if stats else 0, N, groups,
                 1 if cls_bias is not None else 0, 0, 0, 0)
    y = torch.empty((B, Ho, Wo, N), dtype=out_dtype, device=x.device)
    fast = (x.dtype == torch.bfloat16 and out_dtype == torch.bfloat16 and bias is None and act == ACT_NONE
            and N % 8 == 0 and lib.s2i_conv_bf16_eligible(ctypes.byref(d)))
    if EXEC_LOG is not None:
        T = _TAPS[kind]
        _log_exec("conv%s k%d %s[%d,%d,%d,%d]->%d" % ("16" if fast else "", kind, "T" if wmode else "", B, H, W, Cx, N),
                  B * Ho * Wo, N, T * Cx, x.numel(), T * Cx * N * (4 if kind == TCONV_K4S2 else 1), y.numel(),
                  x.element_size(), y.element_size())
    part, nparts = None, 0
    if fast:
        if stats:
            nparts = lib.s2i_conv_bf16_stat_parts(ctypes.byref(d))
            if nparts <= 0:
                check(1, "s2i_conv_bf16_stat_parts")
            part = torch.empty((2, nparts, N), dtype=torch.float32, device=x.device)
        wb = bf16_weight(packed, d, w_offset)
        ws = _ws.get(lib.s2i_conv_bf16_workspace_bytes(ctypes.byref(d)), x.device)
        check(lib.s2i_conv_forward_bf16(ctypes.byref(d), ptr(x), ptr(wb), ptr(cls_bias), ptr(y), ptr(part), ptr(ws),
                                        ws.numel() * 4, stream()), "s2i_conv_forward_bf16")
        return y, part, nparts
    if stats:
        nparts = lib.s2i_conv_stat_parts(ctypes.byref(d))
        if nparts <= 0:
            check(1, "s2i_conv_stat_parts")
        part = torch.empty((2, nparts, N), dtype=torch.float32, device=x.device)
    ws = _ws.get(lib.s2i_conv_workspace_bytes(ctypes.byref(d)), x.device)
    check(lib.s2i_conv_forward_dt(ctypes.byref(d), ptr(x), _dt(x), None, ptr(packed) + 4 * int(w_offset), ptr(bias),
                                  ptr(cls_bias), ptr(y), _dt(y), ptr(part), ptr(ws), ws.numel() * 4, stream()),
          "s2i_conv_forward_dt")
    return y, part, nparts


def wgrad_any(kind, a, g, grad_shape, *, swap=0, fold=0, out=None, accumulate=False, i_off=0, I_total=0):
    """Weight gradient from operands of either activation dtype (no broadcast vector) into an fp32 OIHW tensor."""
    lib = _lib_ready()
    B, H, W, Ca = a.shape
    N = g.shape[-1]
    if len(grad_shape) == 2:
        O, I, KH, KW = grad_shape[0], grad_shape[1], 1, 1
    else:
        O, I, KH, KW = grad_shape
    d = WgradDesc(kind, B, H, W, Ca, 0, N, N, swap, fold, O, I, KH, KW, 1 if accumulate else 0, i_off, I_total)
    if EXEC_LOG is not None:
        Ho, Wo = _geom(kind, H, W)
        _log_exec("wgrad%s k%d a[%d,%d,%d,%d] g%d" % ("16" if a.dtype == g.dtype == torch.bfloat16 else "", kind, B, H, W,
                                                      Ca, N), B * Ho * Wo, N, _TAPS[kind] * Ca, a.numel(), g.numel(),
                  _TAPS[kind] * Ca * N, a.element_size(), 4)
    if out is None:
        full = grad_shape if not I_total else (O, I_total, KH, KW)
        out = torch.empty(full, dtype=torch.float32, device=a.device)
    wsb = lib.s2i_wgrad_workspace_bytes_dt(ctypes.byref(d), _dt(a), _dt(g))
    if wsb == 0:
        check(1, "s2i_wgrad_workspace_bytes_dt")
    ws = _ws.get(wsb, a.device)
    check(lib.s2i_conv_wgrad_dt(ctypes.byref(d), ptr(a), _dt(a), None, ptr(g), _dt(g), ptr(out), ptr(ws), ws.numel() * 4,
                                stream()), "s2i_conv_wgrad_dt")
    return out


# ---- executed-work accounting ------------------------------------------------------------------------
# bench.py / tools set EXEC_LOG to a list for the duration of ONE step: every matrix-product launch appends
# (what, flops, bytes) with the multiply-add count the kernels really execute (2*M*N*K of the launched GEMM: up-blocks at
# 4 taps per output parity, folded c_code channels and skipped weight gradients NOT counted) and the operand + result
# bytes of the launch.  None (default) costs one comparison per launch.
EXEC_LOG = None
_TAPS = {CONV_K1: 1, CONV_K3S1: 9, CONV_K4S2: 16, TCONV_K4S2: 4}


def _log_exec(what, M, N, K, in_elems, w_elems, out_elems, esize_in=4, esize_out=4):
    if EXEC_LOG is not None:
        EXEC_LOG.append((what, 2.0 * M * N * K, in_elems * esize_in + w_elems * esize_in + out_elems * esize_out, M, N, K))


# ---- raw kernels ------------------------------------------------------------------------------------
def _geom(kind, H, W):
    if kind == CONV_K4S2:
        return H // 2, W // 2
    if kind == TCONV_K4S2:
        return 2 * H, 2 * W
    return H, W


# rows per block of the fp32 matrix kernel: 0 = the planner's choice; tools / tests set 96 or 128 to force a variant
TILE_ROWS = 0


# ---- deferred BatchNorm + activation ("apply-on-load") ------------------------------------------------------------------
# A block whose ONLY consumer is the next convolution can skip its normalise + LeakyReLU pass: ConvBnAct(..., defer=True)
# returns its activated tensor UNWRITTEN, tagged with the raw conv output y and the coefficient table; a consuming
# ConvBnAct gathers y through s2i_conv_forward_in (and s2i_conv_wgrad_in in its backward), which apply scale / shift /
# LeakyReLU while they stage the operand -- the activated tensor is never written or read.  Anything else that touches a
# tagged tensor goes through real(), which runs the skipped pass into the tensor's storage first.  Callers opt in per block
# (model.py: the discriminator towers' chains); fp32 tensors, LeakyReLU, training mode.
# MEASURED (round 3, one MI355X, batch 24): bit-for-rounding equal to the separate pass (tests/test_parity_gpu.py::
# test_apply_on_load_equals_the_separate_activation_pass) and SLOWER: 32.5 vs 31.65 ms per step.  The fp32 matrix kernels
# are bound by the matrix pipe, and an im2col gather sees every input element once per TAP: the fma + LeakyReLU + padding
# select run 9 - 16 times per element in the staging path between two barriers, which costs the 36 fused launches more
# (+7 %) than the 49 removed passes took (0.35 ms of HBM-bound kernels that overlapped other streams anyway).  Off by default
# (S2I_DEFER_ACT=1 turns it on); DESIGN.md section 13 has the byte accounting for the other block types.
DEFER_ACT = os.environ.get("S2I_DEFER_ACT", "0") == "1"


class _Lazy:
    __slots__ = ("y", "coef", "act", "groups", "done")

    def __init__(self, y, coef, act, groups):
        self.y, self.coef, self.act, self.groups, self.done = y, coef, act, groups, False


def real(x):
    """The tensor with its deferred BatchNorm + activation pass executed (no-op for ordinary tensors)."""
    lz = getattr(x, '_s2i_lazy', None)
    if lz is not None and not lz.done:
        lib = _lib_ready()
        C = lz.y.shape[-1]
        M = lz.y.numel() // C
        check(lib.s2i_bn_act_forward_dt(_dt(lz.y), ptr(lz.y), M, lz.groups, C, ptr(lz.coef), lz.act, None, ptr(x),
                                        stream()), "s2i_bn_act_forward")
        lz.done = True
    return x


def _pending(x):
    lz = getattr(x, '_s2i_lazy', None)
    return lz if (lz is not None and not lz.done) else None


def conv_raw(kind, x, cvec, packed, N, *, wmode=0, flip=0, wR, ldw, bias=None, act=ACT_NONE, stats=False, groups=1,
             w_offset=0, cls_bias=None, conv1d=None, in_src=None):
    """x: [B,H,W,Cx] contiguous NHWC.  Returns (y [B,Ho,Wo,N], part or None, nparts).
    w_offset (floats) skips leading weight rows (the c_code rows of a jointConv); cls_bias [B][9][N]
    adds their pre-reduced contribution per border class."""
    lib = _lib_ready()
    B, H, W, Cx = x.shape
    Cc = 0 if cvec is None else cvec.shape[1]
    Ho, Wo = _geom(kind, H, W)
    kw1, st1, pd1 = conv1d if conv1d is not None else (0, 0, 0)
    if conv1d is not None:
        Ho, Wo = H, (W + 2 * pd1 - kw1) // st1 + 1
    d = ConvDesc(kind, B, H, W, Cx, Cc, N, wmode, flip, wR, ldw, act, 1 if stats else 0, N, groups,
                 1 if cls_bias is not None else 0, kw1, st1, pd1, 128 if MATH_PLANES else TILE_ROWS,
                 in_src.act if in_src is not None else 0, in_src.groups if in_src is not None else 0)
    y = torch.empty((B, Ho, Wo, N), dtype=torch.float32, device=x.device)
    if EXEC_LOG is not None:
        T = kw1 if conv1d is not None else _TAPS[kind]
        # TCONV: Ho x Wo is the full-resolution output grid and every output pixel sees 4 taps
        _log_exec("conv k%d %s[%d,%d,%d,%d+%d]->%d" % (kind, "T" if wmode else "", B, H, W, Cx, Cc, N), B * Ho * Wo, N,
                  T * (Cx + Cc), x.numel(), T * (Cx + Cc) * N * (4 if kind == TCONV_K4S2 else 1), y.numel())
    part, nparts = None, 0
    if stats:
        nparts = lib.s2i_conv_stat_parts(ctypes.byref(d))
        if nparts <= 0:
            check(1, "s2i_conv_stat_parts")
        part = torch.empty((2, nparts, N), dtype=torch.float32, device=x.device)
    wsb = lib.s2i_conv_workspace_bytes(ctypes.byref(d))
    ws = _ws.get(wsb, x.device)
    if (MATH_PLANES and in_src is None and w_offset == 0 and packed.dim() == 3
            and getattr(packed, '_s2i_gen', None) is not None and lib.s2i_conv_split_eligible(ctypes.byref(d))):
        transpose = wmode == 0
        np_, kp = (packed.shape[2], packed.shape[1]) if transpose else (packed.shape[1], packed.shape[2])
        if kp % 8 == 0:
            wsp = split_weight(packed, MATH_PLANES, transpose)
            check(lib.s2i_conv_forward_split(ctypes.byref(d), ptr(x), ptr(cvec), ptr(wsp), MATH_PLANES, np_, kp, ptr(bias),
                                             ptr(cls_bias), ptr(y), ptr(part), ptr(ws), ws.numel() * 4, stream()),
                  "s2i_conv_forward_split")
            return y, part, nparts
    wp = ptr(packed) + 4 * int(w_offset)
    if in_src is not None:
        # x is the raw output of the producing block: its BatchNorm + LeakyReLU are applied while the operand is staged
        check(lib.s2i_conv_forward_in(ctypes.byref(d), ptr(x), ptr(in_src.coef), wp, ptr(y), ptr(part), ptr(ws),
                                      ws.numel() * 4, stream()), "s2i_conv_forward_in")
        return y, part, nparts
    check(lib.s2i_conv_forward_cls(ctypes.byref(d), ptr(x), ptr(cvec), wp, ptr(bias), ptr(cls_bias), ptr(y), ptr(part),
                                   ptr(ws), ws.numel() * 4, stream()), "s2i_conv_forward")
    return y, part, nparts


def wgrad_raw(kind, a, cvec, g, grad_shape, *, swap=0, fold=0, out=None, accumulate=False, i_off=0, I_total=0, a_src=None):
    """Weight gradient into an OIHW tensor of shape grad_shape.  a: gathered NHWC, g: plain NHWC.
    With I_total > 0 only input channels [i_off, i_off + I) of a wider (O, I_total, KH, KW) tensor are written."""
    lib = _lib_ready()
    B, H, W, Ca = a.shape
    Cc = 0 if cvec is None else cvec.shape[1]
    N = g.shape[-1]
    if len(grad_shape) == 2:
        O, I, KH, KW = grad_shape[0], grad_shape[1], 1, 1
    else:
        O, I, KH, KW = grad_shape
    d = WgradDesc(kind, B, H, W, Ca, Cc, N, N, swap, fold, O, I, KH, KW, 1 if accumulate else 0, i_off, I_total,
                  a_src.act if a_src is not None else 0, a_src.groups if a_src is not None else 0)
    if EXEC_LOG is not None:
        Ho, Wo = _geom(kind, H, W)
        _log_exec("wgrad k%d a[%d,%d,%d,%d+%d] g%d" % (kind, B, H, W, Ca, Cc, N), B * Ho * Wo, N, _TAPS[kind] * (Ca + Cc),
                  a.numel(), g.numel(), _TAPS[kind] * (Ca + Cc) * N)
    if out is None:
        full = grad_shape if not I_total else (O, I_total, KH, KW)
        out = torch.empty(full, dtype=torch.float32, device=a.device)
    if a_src is not None:
        # `a` is the producer's raw output; eligibility was checked by the caller (s2i_conv_wgrad_in_eligible)
        wsb = lib.s2i_wgrad_workspace_bytes(ctypes.byref(d))
        if wsb == 0:
            check(1, "s2i_wgrad_workspace_bytes")
        ws = _ws.get(wsb, a.device)
        check(lib.s2i_conv_wgrad_in(ctypes.byref(d), ptr(a), ptr(a_src.coef), ptr(g), ptr(out), ptr(ws), ws.numel() * 4,
                                    stream()), "s2i_conv_wgrad_in")
        return out
    if MATH_PLANES:
        wsb = lib.s2i_wgrad_workspace_bytes_split(ctypes.byref(d), MATH_PLANES)
        if wsb == 0:
            check(1, "s2i_wgrad_workspace_bytes_split")
        ws = _ws.get(wsb, a.device)
        check(lib.s2i_conv_wgrad_split(ctypes.byref(d), MATH_PLANES, ptr(a), ptr(cvec), ptr(g), ptr(out), ptr(ws),
                                       ws.numel() * 4, stream()), "s2i_conv_wgrad_split")
        return out
    wsb = lib.s2i_wgrad_workspace_bytes(ctypes.byref(d))
    if wsb == 0:
        check(1, "s2i_wgrad_workspace_bytes")
    ws = _ws.get(wsb, a.device)
    check(lib.s2i_conv_wgrad(ctypes.byref(d), ptr(a), ptr(cvec), ptr(g), ptr(out), ptr(ws), ws.numel() * 4,
                             stream()), "s2i_conv_wgrad")
    return out


def _rows_view(t):
    """(tensor usable by the kernels, row stride) for a NHWC gradient that may be a channel slice."""
    C = t.shape[-1]
    if t.is_contiguous():
        return t, C
    st = t.stride()
    if t.dim() >= 2 and st[-1] == 1 and st[-2] % 4 == 0 and t.data_ptr() % (4 * t.element_size()) == 0:
        ld = st[-2]
        ok = True
        expect = ld
        for dim in range(t.dim() - 2, -1, -1):
            if t.shape[dim] != 1 and st[dim] != expect:
                ok = False
                break
            expect *= t.shape[dim]
        if ok:
            return t, ld
    return t.contiguous(), C


def _num_parts(M):
    """Row chunks of the BatchNorm backward reduction: enough blocks to keep the HBM pipes full on the large maps
    (1024 / 2048 / 4096 chunks measured on the bf16 step: no difference)."""
    return int(max(1, min(1024, M // 64)))


# ---- conv + BatchNorm + activation ---------------------------------------------------------------------
_KIND = {"k1": CONV_K1, "k3s1": CONV_K3S1, "k4s2": CONV_K4S2, "up": TCONV_K4S2}


_DGRAD = {"k3s1": (CONV_K3S1, 1), "k4s2": (TCONV_K4S2, 0), "up": (CONV_K4S2, 0), "k1": (CONV_K1, 0)}


def _dgrad(kind_name, dy, w, packed, n_in, out_dtype=None):
    """Input gradient of the conv `kind_name` given dy (NHWC) -> [B,H,W,n_in]."""
    Op = packed.shape[2]
    if dy.shape[-1] != Op:
        raise _lib.S2IError("dgrad: dy has %d channels, packed weight %d" % (dy.shape[-1], Op))
    if out_dtype is not None:     # bf16 activation mode: either dtype on either side
        kind, flip = _DGRAD[kind_name]
        return conv_any(kind, dy, packed, n_in, wmode=1, flip=flip, out_dtype=out_dtype)[0]
    if kind_name == "k3s1":
        y, _, _ = conv_raw(CONV_K3S1, dy, None, packed, n_in, wmode=1, flip=1, wR=packed.shape[1], ldw=Op)
    elif kind_name == "k4s2":
        y, _, _ = conv_raw(TCONV_K4S2, dy, None, packed, n_in, wmode=1, wR=packed.shape[1], ldw=Op)
    elif kind_name == "up":
        y, _, _ = conv_raw(CONV_K4S2, dy, None, packed, n_in, wmode=1, wR=packed.shape[1], ldw=Op)
    else:
        y, _, _ = conv_raw(CONV_K1, dy, None, packed, n_in, wmode=1, wR=packed.shape[1], ldw=Op)
    return y


def _wgrad(kind_name, x, cvec, dy, weight, a_src=None):
    """Weight gradient; accumulated into weight.grad in place (returns None) in direct mode.  a_src: x is the RAW output of
    the producing block (deferred BatchNorm + LeakyReLU, applied by the gather)."""
    out, acc = (weight.grad, True) if _direct(weight) else (None, False)

    def run():
        if a_src is not None:
            return wgrad_raw(_KIND[kind_name], x, None, dy, tuple(weight.shape), out=out, accumulate=acc, a_src=a_src)
        if x.dtype == torch.bfloat16 or dy.dtype == torch.bfloat16:
            if cvec is not None:
                raise _lib.S2IError("wgrad: bf16 operands carry their broadcast vector materialised")
            if kind_name == "up":
                return wgrad_any(CONV_K4S2, dy, x, tuple(weight.shape), swap=1, fold=1, out=out, accumulate=acc)
            g = dy
            if (x.dtype == torch.bfloat16 and dy.dtype == torch.float32 and dy.shape[-1] <= 4
                    and x.shape[-1] % 8 == 0):
                # GET_IMAGE_G's weight gradient (bf16 features x fp32 NHWC4 image gradient): the <= 4 channel stream kernel
                # ran at a tenth of the HBM rate (0.32 ms at 256 px); padded to 8 bf16 channels the image gradient goes
                # through the bf16-MFMA weight-gradient kernel like every other layer's
                g = torch.zeros(dy.shape[:-1] + (8,), dtype=torch.bfloat16, device=dy.device)
                g[..., :dy.shape[-1]] = dy
            return wgrad_any(_KIND[kind_name], x, g, tuple(weight.shape), out=out, accumulate=acc)
        if kind_name == "up":
            return wgrad_raw(CONV_K4S2, dy, None, x, tuple(weight.shape), swap=1, fold=1, out=out, accumulate=acc)
        return wgrad_raw(_KIND[kind_name], x, cvec, dy, tuple(weight.shape), out=out, accumulate=acc)

    dw = run()
    return None if acc else dw


def _split_input_grad(dx_full, Cc):
    """dx of the stored channels and the gradient of the broadcast vector."""
    if Cc == 0:
        return dx_full, None
    lib = _lib_ready()
    B, H, W, Ca = dx_full.shape
    if H * W == 1:
        dc = dx_full.reshape(B, Ca)[:, :Cc].float()
    else:
        dc = torch.empty((B, Cc), dtype=torch.float32, device=dx_full.device)
        wsb = lib.s2i_spatial_sum_workspace_bytes(B, H * W, Cc)
        ws = _ws.get(wsb, dx_full.device)
        check(lib.s2i_spatial_sum_dt(_dt(dx_full), ptr(dx_full), Ca, B, H * W, Cc, ptr(dc), ptr(ws), ws.numel() * 4,
                                     stream()), "s2i_spatial_sum")
    return dx_full[..., Cc:], dc


def _factor_cvec(kind_name, cvec, x):
    """Spatially constant channels of a 3x3 conv are folded into a per-image, per-border-class bias (and
    their gradients taken from border sums of dY) instead of being convolved: on G's jointConv layers
    (model.py:268, 272-279) they are 128 of 192 / 160 input channels."""
    return cvec is not None and kind_name == "k3s1" and x.shape[1] >= 16 and x.shape[2] >= 16 and cvec.shape[1] % 4 == 0


def _tap_sums(dy):
    lib = _lib_ready()
    B, H, W, C = dy.shape
    out = torch.empty((B, 9, C), dtype=torch.float32, device=dy.device)
    wsb = lib.s2i_border_sums_workspace_bytes(B, H, W, C)
    ws = _ws.get(wsb, dy.device)
    check(lib.s2i_tap_sums_dt(_dt(dy), ptr(dy), B, H, W, C, ptr(out), ptr(ws), ws.numel() * 4, stream()), "s2i_tap_sums")
    return out


class ConvBnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cvec, weight, gamma, beta, residual, kind_name, act, bn_buffers, training, groups=1, defer=False):
        lib = _lib_ready()
        # input produced by a block that deferred its BatchNorm + LeakyReLU: gather its raw output instead (apply-on-load)
        in_src = _pending(x)
        if in_src is not None:
            if (cvec is None and residual is None and kind_name in ("k3s1", "k4s2") and not MATH_PLANES and training
                    and in_src.y.dtype == torch.float32 and x.shape[-1] % 32 == 0 and weight.shape[0] > 4
                    and in_src.groups == (groups if groups else 1)):
                x = in_src.y
            else:
                real(x)
                in_src = None
        x = x.contiguous()
        if cvec is not None:
            cvec = cvec.contiguous()
        kind = _KIND[kind_name]
        packed = packed_weight(weight, PACK_UPFOLD if kind_name == "up" else PACK_PLAIN)
        Cout = weight.shape[0]
        factored = _factor_cvec(kind_name, cvec, x)
        # bf16 activation mode: every spatial block stores its raw conv output and its result as bf16 (the fc of
        # INIT_STAGE_G, kind k1, stays an fp32 island)
        bf = (ACT_BF16 and kind_name != "k1") or x.dtype == torch.bfloat16
        if bf and in_src is not None:
            raise _lib.S2IError("ConvBnAct: apply-on-load input in the bf16 activation mode")
        adt = torch.bfloat16 if bf else torch.float32
        cat_cc = 0
        if bf and cvec is not None and not factored:
            # D's jointConv on 4x4 maps: the (c_code, h) concat of model.py:434 is materialised (a few MB) in bf16
            B_, H_, W_, _ = x.shape
            cat_cc = cvec.shape[1]
            x = torch.cat((cast(cvec, adt).view(B_, 1, 1, cat_cc).expand(B_, H_, W_, cat_cc), cast(x, adt)), 3).contiguous()
            cvec_used = None
        else:
            cvec_used = cvec
        if factored:
            B, Cc = cvec.shape
            Ip, Op = packed.shape[1], packed.shape[2]
            table = torch.empty((B, 9, Cout), dtype=torch.float32, device=x.device)
            ws = _ws.get(B * 9 * Cout * 4, x.device)
            check(lib.s2i_cvec_bias_table(ptr(cvec), ptr(packed), B, Cc, Ip, Op, Cout, ptr(table), ptr(ws),
                                          ws.numel() * 4, stream()), "s2i_cvec_bias_table")
            if bf:
                y, part, nparts = conv_any(kind, x, packed, Cout, stats=training, groups=groups, w_offset=Cc * Op,
                                           cls_bias=table, out_dtype=adt)
            else:
                y, part, nparts = conv_raw(kind, x, None, packed, Cout, wR=Ip, ldw=Op, stats=training, groups=groups,
                                           w_offset=Cc * Op, cls_bias=table)
        elif bf:
            y, part, nparts = conv_any(kind, x, packed, Cout, stats=training, groups=groups, out_dtype=adt)
        else:
            y, part, nparts = conv_raw(kind, x, cvec, packed, Cout, wR=packed.shape[1], ldw=packed.shape[2],
                                       stats=training, groups=groups, in_src=in_src)
        M = y.numel() // Cout
        if not training:
            groups = 1
        coef = torch.empty((groups, 4, Cout), dtype=torch.float32, device=x.device)
        rm, rv, nbt = bn_buffers
        if training:
            check(lib.s2i_bn_finalize(ptr(part), nparts, groups, Cout, M // groups, ptr(gamma), ptr(beta), ptr(rm),
                                      ptr(rv), ptr(nbt), BN_MOMENTUM, BN_EPS, ptr(coef), stream()), "s2i_bn_finalize")
        else:
            check(lib.s2i_bn_eval_coeffs(Cout, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), BN_EPS, ptr(coef), stream()),
                  "s2i_bn_eval_coeffs")
        Cact = Cout // 2 if act == ACT_GLU else Cout
        out = torch.empty(y.shape[:-1] + (Cact,), dtype=adt, device=x.device)
        if residual is not None:
            residual = cast(real(residual), adt).contiguous()
        if defer and DEFER_ACT and training and act == ACT_LRELU and residual is None and not bf and not MATH_PLANES:
            # the caller guarantees a single convolution consumer: `out` stays unwritten until real() or never
            out._s2i_lazy = _Lazy(y, coef, act, groups)
        else:
            check(lib.s2i_bn_act_forward_dt(_dt(y), ptr(y), M, groups, Cout, ptr(coef), act, ptr(residual), ptr(out),
                                            stream()), "s2i_bn_act_forward")
        ctx.save_for_backward(x, cvec_used, weight, gamma, y, coef, None if in_src is None else in_src.coef)
        ctx.in_src = None if in_src is None else (in_src.act, in_src.groups)
        ctx.beta_ref = beta
        ctx.kind_name, ctx.act, ctx.training, ctx.has_res = kind_name, act, training, residual is not None
        ctx.groups = groups
        ctx.factored = factored
        ctx.bf, ctx.cat_cc = bf, cat_cc
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib_ready()
        x, cvec, weight, gamma, y, coef, in_coef = ctx.saved_tensors
        if not ctx.training:
            raise _lib.S2IError("ConvBnAct.backward: eval-mode BatchNorm has no backward on this path")
        a_src = None
        if ctx.in_src is not None:
            # x is the producer's RAW output (apply-on-load forward): the weight gradient gathers it the same way where its
            # plan allows, otherwise the activated operand is computed now
            a_src = _Lazy(x, in_coef, ctx.in_src[0], ctx.in_src[1])
        Cout = weight.shape[0]
        M = y.numel() // Cout
        if dout.dtype != y.dtype:
            dout = cast(dout, y.dtype)
        dout_k, ldd = _rows_view(dout)
        G = ctx.groups
        nparts = _num_parts(M // G) * G
        part = torch.empty((2, nparts, Cout), dtype=torch.float32, device=y.device)
        check(lib.s2i_bn_act_bwd_reduce_dt(_dt(y), ptr(y), ptr(dout_k), ldd, M, G, Cout, ptr(coef), ctx.act, ptr(part),
                                           nparts, stream()), "s2i_bn_act_bwd_reduce")
        beta = ctx.beta_ref
        need_gb = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        direct_bn = need_gb and _direct(gamma) and _direct(beta)
        if not need_gb:       # frozen BatchNorm parameters (the discriminators during the G update)
            dgamma = dbeta = None
        else:
            dgamma = gamma.grad if direct_bn else torch.empty_like(gamma)
            dbeta = beta.grad if direct_bn else torch.empty_like(gamma)
        red2 = torch.empty((G, 2, Cout), dtype=torch.float32, device=y.device)
        check(lib.s2i_bn_bwd_finalize(ptr(part), nparts, G, Cout, M // G, ptr(dgamma), ptr(dbeta),
                                      1 if direct_bn else 0, ptr(red2), stream()), "s2i_bn_bwd_finalize")
        if direct_bn:
            dgamma = dbeta = None
        dy = torch.empty_like(y)
        check(lib.s2i_bn_act_bwd_apply_dt(_dt(y), ptr(y), ptr(dout_k), ldd, M, G, Cout, ptr(coef), ptr(red2), ctx.act,
                                          ptr(dy), stream()), "s2i_bn_act_bwd_apply")
        need_x, need_c, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dx = dc = dw = None
        Cc = 0 if cvec is None else cvec.shape[1]
        bf = ctx.bf
        if bf and not ctx.factored:
            # bf16 activation mode; a broadcast vector (ctx.cat_cc channels) was concatenated in front of x in the forward
            cc = ctx.cat_cc
            if need_x or (need_c and cc):
                packed = packed_weight(weight, PACK_UPFOLD if ctx.kind_name == "up" else PACK_PLAIN)
                dx_full = _dgrad(ctx.kind_name, dy, weight, packed, x.shape[-1], out_dtype=x.dtype)
                dx, dc = _split_input_grad(dx_full, cc)
                if not need_x:
                    dx = None
                if not need_c:
                    dc = None
            if need_w:
                dw = _wgrad(ctx.kind_name, x, None, dy, weight)
            dres = dout if ctx.has_res else None
            return dx, dc, dw, dgamma, dbeta, dres, None, None, None, None, None, None
        if ctx.factored:
            packed = packed_weight(weight, PACK_PLAIN)
            Ip, Op = packed.shape[1], packed.shape[2]
            B, Cx = x.shape[0], x.shape[-1]
            if need_x:
                if bf:
                    dx = conv_any(CONV_K3S1, dy, packed, Cx, wmode=1, flip=1, w_offset=Cc * Op, out_dtype=x.dtype)[0]
                else:
                    dx, _, _ = conv_raw(CONV_K3S1, dy, None, packed, Cx, wmode=1, flip=1, wR=Ip, ldw=Op, w_offset=Cc * Op)
            if need_c or need_w:
                tapsum = _tap_sums(dy)
                direct = need_w and _direct(weight)
                if need_w:
                    dw = weight.grad if direct else torch.empty(tuple(weight.shape), dtype=torch.float32, device=x.device)
                    if bf:
                        wgrad_any(CONV_K3S1, x, dy, (Cout, Cx, 3, 3), out=dw, accumulate=direct, i_off=Cc, I_total=Cc + Cx)
                    else:
                        wgrad_raw(CONV_K3S1, x, None, dy, (Cout, Cx, 3, 3), out=dw, accumulate=direct, i_off=Cc,
                                  I_total=Cc + Cx)
                if need_c:
                    dc = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
                check(lib.s2i_cvec_grads(ptr(cvec), ptr(packed), ptr(tapsum), B, Cc, Ip, Op, Cout, Cout, Cc + Cx, ptr(dc),
                                         ptr(dw) if need_w else None, 1 if direct else 0, stream()), "s2i_cvec_grads")
                if direct:
                    dw = None
        elif need_x or (need_c and cvec is not None):
            packed = packed_weight(weight, PACK_UPFOLD if ctx.kind_name == "up" else PACK_PLAIN)
            dx_full = _dgrad(ctx.kind_name, dy, weight, packed, x.shape[-1] + Cc)
            dx, dc = _split_input_grad(dx_full, Cc)
            if not need_x:
                dx = None
        if need_w and not ctx.factored:
            if a_src is not None:
                B_, H_, W_, Ca_ = x.shape
                O_, I_, KH_, KW_ = weight.shape
                dd = WgradDesc(_KIND[ctx.kind_name], B_, H_, W_, Ca_, 0, dy.shape[-1], dy.shape[-1], 0, 0, O_, I_, KH_, KW_, 0,
                               0, 0, a_src.act, a_src.groups)
                if lib.s2i_conv_wgrad_in_eligible(ctypes.byref(dd)):
                    dw = _wgrad(ctx.kind_name, x, None, dy, weight, a_src=a_src)
                else:
                    xa = torch.empty_like(x)
                    check(lib.s2i_bn_act_forward_dt(_dt(x), ptr(x), x.numel() // Ca_, a_src.groups, Ca_, ptr(in_coef),
                                                    a_src.act, None, ptr(xa), stream()), "s2i_bn_act_forward")
                    dw = _wgrad(ctx.kind_name, xa, None, dy, weight)
            else:
                dw = _wgrad(ctx.kind_name, x, cvec, dy, weight)
        dres = dout if ctx.has_res else None
        return dx, dc, dw, dgamma, dbeta, dres, None, None, None, None, None, None


class ConvAct(torch.autograd.Function):
    """conv (+bias) + LeakyReLU / tanh / none, no BatchNorm."""

    @staticmethod
    def forward(ctx, x, weight, bias, kind_name, act, n_out):
        x = real(x).contiguous()
        kind = _KIND[kind_name]
        packed = packed_weight(weight, PACK_PLAIN)
        if (ACT_BF16 and kind_name != "k1") or x.dtype == torch.bfloat16:
            # bf16 activation mode: feature maps (>= 8 channels) are bf16, NHWC4 images stay fp32
            odt = torch.bfloat16 if n_out >= 8 else torch.float32
            out = conv_any(kind, x, packed, n_out, bias=bias, act=act, out_dtype=odt)[0]
        else:
            out, _, _ = conv_raw(kind, x, None, packed, n_out, wR=packed.shape[1], ldw=packed.shape[2], bias=bias, act=act)
        ctx.save_for_backward(x, weight, out if act != ACT_NONE else None)
        ctx.kind_name, ctx.act, ctx.has_bias, ctx.n_out = kind_name, act, bias is not None, n_out
        ctx.bias_ref = bias
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib_ready()
        x, weight, out = ctx.saved_tensors
        N = ctx.n_out
        M = dout.numel() // N
        if out is not None and dout.dtype != out.dtype:
            dout = cast(dout, out.dtype)
        dout_k, ldd = _rows_view(dout)
        if ctx.act != ACT_NONE:
            dy = torch.empty(dout.shape, dtype=dout.dtype, device=dout.device)
            check(lib.s2i_act_backward_dt(_dt(out), ptr(out), ptr(dout_k), ldd, M, N, ctx.act, ptr(dy), stream()),
                  "s2i_act_backward")
        else:
            dy = dout_k if ldd == N else dout_k.contiguous()
        dx = dw = db = None
        mixed = x.dtype == torch.bfloat16 or dy.dtype == torch.bfloat16
        if ctx.needs_input_grad[0]:
            packed = packed_weight(weight, PACK_PLAIN)
            dx = _dgrad(ctx.kind_name, dy, weight, packed, x.shape[-1], out_dtype=x.dtype if mixed else None)
        if ctx.needs_input_grad[1]:
            dw = _wgrad(ctx.kind_name, x, None, dy, weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            part = torch.empty((2, 1, N), dtype=torch.float32, device=dy.device)
            check(lib.s2i_colstats(ptr(cast(dy, torch.float32)), M, N, N, ptr(part), 1, stream()), "s2i_colstats")
            bias = ctx.bias_ref
            O = weight.shape[0]
            if _direct(bias):      # straight into the flat gradient buffer: no temporary, no AccumulateGrad add
                check(lib.s2i_axpby(ptr(bias.grad), ptr(part), O, 1.0, 1.0, stream()), "s2i_axpby")
            else:
                db = torch.empty((O,), dtype=torch.float32, device=dy.device)
                check(lib.s2i_axpby(ptr(db), ptr(part), O, 1.0, 0.0, stream()), "s2i_axpby")
        return dx, dw, db, None, None, None


# ---- CA_NET pieces -----------------------------------------------------------------------------------------
class Glu2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib_ready()
        x = x.contiguous()
        M, C = x.shape
        out = torch.empty((M, C // 2), dtype=torch.float32, device=x.device)
        check(lib.s2i_glu_forward(ptr(x), M, C, ptr(out), stream()), "s2i_glu_forward")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib_ready()
        (x,) = ctx.saved_tensors
        M, C = x.shape
        dx = torch.empty_like(x)
        check(lib.s2i_glu_backward(ptr(x), ptr(dout.contiguous()), M, C, ptr(dx), stream()), "s2i_glu_backward")
        return dx


class Reparam(torch.autograd.Function):
    """h = [mu | logvar] (B, 2E), eps (B, E) -> c = eps*exp(0.5*logvar) + mu  (model.py:188-195)."""

    @staticmethod
    def forward(ctx, h, eps):
        lib = _lib_ready()
        h = h.contiguous()
        eps = eps.contiguous()
        B, E2 = h.shape
        c = torch.empty((B, E2 // 2), dtype=torch.float32, device=h.device)
        check(lib.s2i_reparam_forward(ptr(h), ptr(eps), B, E2 // 2, ptr(c), stream()), "s2i_reparam_forward")
        ctx.save_for_backward(h, eps)
        return c

    @staticmethod
    def backward(ctx, dc):
        lib = _lib_ready()
        h, eps = ctx.saved_tensors
        B, E2 = h.shape
        dh = torch.empty_like(h)
        check(lib.s2i_reparam_backward(ptr(h), ptr(eps), ptr(dc.contiguous()), None, None, B, E2 // 2, ptr(dh),
                                       stream()), "s2i_reparam_backward")
        return dh, None


class KLLoss(torch.autograd.Function):
    """KL_loss of trainer.py:54-58 on (mu, logvar), each (B, E), possibly column slices of one tensor."""

    @staticmethod
    def forward(ctx, mu, logvar):
        lib = _lib_ready()
        if mu.stride(1) != 1:
            mu = mu.contiguous()
        if logvar.stride(1) != 1:
            logvar = logvar.contiguous()
        B, E = mu.shape
        kl = torch.empty((), dtype=torch.float32, device=mu.device)
        check(lib.s2i_kl_forward(ptr(mu), mu.stride(0), ptr(logvar), logvar.stride(0), B, E, ptr(kl), stream()),
              "s2i_kl_forward")
        ctx.save_for_backward(mu, logvar)
        return kl

    @staticmethod
    def backward(ctx, g):
        lib = _lib_ready()
        mu, logvar = ctx.saved_tensors
        B, E = mu.shape
        dmu = torch.empty((B, E), dtype=torch.float32, device=mu.device)
        dlv = torch.empty((B, E), dtype=torch.float32, device=mu.device)
        check(lib.s2i_kl_backward(ptr(mu), mu.stride(0), ptr(logvar), logvar.stride(0), B, E, ptr(g.contiguous()),
                                  ptr(dmu), ptr(dlv), stream()), "s2i_kl_backward")
        return dmu, dlv


# ---- D heads and losses ---------------------------------------------------------------------------------------
class LogitHead(torch.autograd.Function):
    """Conv2d(C,1,k=4,s=4)+Sigmoid on a (B,4,4,C) NHWC map (model.py:414-422)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib_ready()
        ctx.x_dtype = x.dtype
        x = cast(real(x).contiguous(), torch.float32)   # a (B,4,4,C) map: the heads and the losses are an fp32 island
        B, H, W, C = x.shape
        if H != 4 or W != 4 or tuple(weight.shape) != (1, C, 4, 4):
            raise _lib.S2IError("LogitHead: expects a 4x4 map and a (1,C,4,4) weight")
        prob = torch.empty((B,), dtype=torch.float32, device=x.device)
        check(lib.s2i_logit_forward(ptr(x), ptr(weight.contiguous()), ptr(bias), B, C, ptr(prob), stream()),
              "s2i_logit_forward")
        ctx.save_for_backward(x, weight, prob)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        return prob

    @staticmethod
    def backward(ctx, dprob):
        lib = _lib_ready()
        x, weight, prob = ctx.saved_tensors
        B, _, _, C = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        bias = ctx.bias_ref
        want_w = ctx.needs_input_grad[1]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        direct = want_w and _direct(weight) and (not want_b or _direct(bias))
        if direct:
            dw, db = weight.grad, (bias.grad if want_b else None)
        else:
            dw = torch.empty_like(weight) if want_w else None
            db = torch.empty((1,), dtype=torch.float32, device=x.device) if want_b else None
        check(lib.s2i_logit_backward(ptr(x), ptr(weight.contiguous()), ptr(prob), ptr(dprob.contiguous()), B, C,
                                     ptr(dx), 0, ptr(dw), ptr(db), 1 if direct else 0, stream()), "s2i_logit_backward")
        if direct:
            dw = db = None
        if dx is not None:
            dx = cast(dx, ctx.x_dtype)
        return dx, dw, db


class BCELoss(torch.autograd.Function):
    """weight * nn.BCELoss()(prob, full(target)) (trainer.py:394-409, 439-443)."""

    @staticmethod
    def forward(ctx, prob, target, weight):
        lib = _lib_ready()
        prob = prob.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=prob.device)
        check(lib.s2i_bce_forward(ptr(prob), float(target), prob.numel(), float(weight), ptr(loss), 0, stream()),
              "s2i_bce_forward")
        ctx.save_for_backward(prob)
        ctx.target, ctx.weight = float(target), float(weight)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib_ready()
        (prob,) = ctx.saved_tensors
        dprob = torch.empty_like(prob)
        check(lib.s2i_bce_backward(ptr(prob), ctx.target, prob.numel(), ctx.weight, ptr(g.contiguous()), ptr(dprob),
                                   stream()), "s2i_bce_backward")
        return dprob, None, None


class BCEMulti(torch.autograd.Function):
    """All BCE terms of one discriminator update in one launch: `heads` = H probability vectors of G stacked
    batches (G*B rows each); term (g, h) has target targets[g][h] and weight weights[g][h]
    (trainer.py:394-409 sums six of them)."""

    @staticmethod
    def forward(ctx, targets_dev, weights_dev, G, *heads):
        lib = _lib_ready()
        heads = [h.contiguous() for h in heads]
        H = len(heads)
        B = heads[0].numel() // G
        arr = (ctypes.c_void_p * H)(*[ptr(h) for h in heads])
        loss = torch.empty((), dtype=torch.float32, device=heads[0].device)
        check(lib.s2i_bce_multi_forward(arr, ptr(targets_dev), ptr(weights_dev), G, H, B, ptr(loss), stream()),
              "s2i_bce_multi_forward")
        ctx.save_for_backward(targets_dev, weights_dev, *heads)
        ctx.G = G
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib_ready()
        targets_dev, weights_dev = ctx.saved_tensors[:2]
        heads = ctx.saved_tensors[2:]
        H, G = len(heads), ctx.G
        B = heads[0].numel() // G
        grads = [torch.empty_like(h) for h in heads]
        pin = (ctypes.c_void_p * H)(*[ptr(h) for h in heads])
        pout = (ctypes.c_void_p * H)(*[ptr(t) for t in grads])
        check(lib.s2i_bce_multi_backward(pin, ptr(targets_dev), ptr(weights_dev), G, H, B, ptr(g.contiguous()), pout,
                                         stream()), "s2i_bce_multi_backward")
        return (None, None, None) + tuple(grads)


class ClassAwareLoss(torch.autograd.Function):
    """class_aware_loss of trainer.py:298-311 on features (B, D) and int32 device labels (B,)."""

    @staticmethod
    def forward(ctx, feats, labels):
        lib = _lib_ready()
        feats = feats.contiguous()
        B, D = feats.shape
        x4 = feats.view(B, 1, 1, D)
        # scores = X X^T through the implicit-GEMM kernel: the second operand is X itself, read transposed
        scores, _, _ = conv_raw(CONV_K1, x4, None, feats, B, wmode=1, wR=B, ldw=D)
        loss = torch.empty((1,), dtype=torch.float32, device=feats.device)
        dS = torch.empty((B, B), dtype=torch.float32, device=feats.device)
        check(lib.s2i_cal_loss(ptr(scores), ptr(labels), B, D, ptr(loss), 0, ptr(dS), stream()), "s2i_cal_loss")
        ctx.save_for_backward(feats, dS)
        return loss

    @staticmethod
    def backward(ctx, g):
        feats, dS = ctx.saved_tensors
        B, D = feats.shape
        Bp = _roundup4(B)
        if Bp != B:
            # the gathered operand needs a channel count that is a multiple of 4: zero columns of dS against zero
            # rows of X are exact (ragged last batch of an epoch, trainer.py:543-545: 8855 % 24 = 23 on CUB)
            dS_p = dS.new_zeros((B, Bp))
            dS_p[:, :B] = dS
            feats_p = feats.new_zeros((Bp, D))
            feats_p[:B] = feats
            dS, feats = dS_p, feats_p
        # dX = dS_sym X : K1 conv with gathered operand dS (B x Bp) and weights X used as P[k][n]
        dX, _, _ = conv_raw(CONV_K1, dS.view(B, 1, 1, Bp), None, feats, D, wmode=0, wR=Bp, ldw=D)
        dX = dX.view(B, D)
        lib = _lib_ready()
        # scale by the incoming gradient, read on the device (no host sync)
        check(lib.s2i_scale_dev(ptr(dX), ptr(dX), dX.numel(), ptr(g.contiguous()), stream()), "s2i_scale_dev")
        return dX, None


# ---- layout ---------------------------------------------------------------------------------------------------------
class ToNHWC(torch.autograd.Function):
    """(B,C,H,W) -> (B,H,W,Cp) with zero-filled padding channels."""

    @staticmethod
    def forward(ctx, x, Cp):
        lib = _lib_ready()
        x = x.contiguous()
        B, C, H, W = x.shape
        # bf16 activation mode: feature maps become bf16 here; NHWC4 images stay fp32
        odt = torch.bfloat16 if (ACT_BF16 and Cp >= 8) else torch.float32
        out = torch.empty((B, H, W, Cp), dtype=odt, device=x.device)
        check(lib.s2i_nchw_to_nhwc_dt(_dt(out), ptr(x), ptr(out), B, C, H, W, Cp, stream()), "s2i_nchw_to_nhwc")
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib_ready()
        dk, ld = _rows_view(dout)
        B, H, W, _ = dout.shape
        dx = torch.empty((B, ctx.C, H, W), dtype=torch.float32, device=dout.device)
        check(lib.s2i_nhwc_to_nchw_dt(_dt(dk), ptr(dk), ld, ptr(dx), B, ctx.C, H, W, stream()), "s2i_nhwc_to_nchw")
        return dx, None


class ToNCHW(torch.autograd.Function):
    """(B,H,W,Cp) -> (B,C,H,W), dropping padding channels."""

    @staticmethod
    def forward(ctx, x, C):
        lib = _lib_ready()
        xk, ld = _rows_view(real(x))
        B, H, W, Cp = x.shape
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        check(lib.s2i_nhwc_to_nchw_dt(_dt(xk), ptr(xk), ld, ptr(out), B, C, H, W, stream()), "s2i_nhwc_to_nchw")
        ctx.Cp, ctx.x_dtype = Cp, x.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib_ready()
        dout = dout.contiguous()
        B, C, H, W = dout.shape
        dx = torch.empty((B, H, W, ctx.Cp), dtype=ctx.x_dtype, device=dout.device)
        check(lib.s2i_nchw_to_nhwc_dt(_dt(dx), ptr(dout), ptr(dx), B, C, H, W, ctx.Cp, stream()), "s2i_nchw_to_nhwc")
        return dx, None


# ---- optimiser -----------------------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, lr, beta1, beta2, eps, step=0, step_dev=None, gscale=1.0):
    lib = _lib_ready()
    check(lib.s2i_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, int(step),
                            ptr(step_dev), gscale, stream()), "s2i_adam_step")


def ema_update(avg, p, decay):
    lib = _lib_ready()
    check(lib.s2i_ema_update(ptr(avg), ptr(p), p.numel(), decay, stream()), "s2i_ema_update")


def increment(counter):
    lib = _lib_ready()
    check(lib.s2i_increment(ptr(counter), stream()), "s2i_increment")




# ---- evaluation output ---------------------------------------------------------------------------------------------
def images_to_uint8_hwc(img_nhwc):
    """(B,H,W,C>=3) float NHWC in [-1,1] -> (B,H,W,3) uint8: the reference's save_singleimages conversion
    (trainer.py:676-677) in one kernel, already in the HWC order a PNG encoder wants."""
    lib = _lib_ready()
    t, ld = _rows_view(img_nhwc)
    B, H, W, _ = img_nhwc.shape
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=img_nhwc.device)
    check(lib.s2i_image_to_u8(ptr(t), ld, ptr(out), B * H * W, stream()), "s2i_image_to_u8")
    return out


# ---- data edge -----------------------------------------------------------------------------------------------------
def images_from_uint8_hwc(u8):
    """(B,H,W,3) uint8 RGB on the device -> (B,3,H,W) float in [-1,1]: the reference's per-sample
    ToTensor + Normalize(0.5, 0.5) (datasets.py:440-442) applied to the collated batch on the GPU, so the
    host->device copy carries 1 byte per sample instead of 4."""
    lib = _lib_ready()
    assert u8.dtype == torch.uint8 and u8.dim() == 4 and u8.shape[-1] == 3 and u8.is_contiguous()
    B, H, W, _ = u8.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=u8.device)
    check(lib.s2i_u8_to_image(ptr(u8), ptr(out), B, H, W, stream()), "s2i_u8_to_image")
    return out
