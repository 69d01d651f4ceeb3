"""Native fp32 MFMA vs split-bf16 (2 / 3 planes) convolution GEMMs: time and accuracy (dev tool, GPU box only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import ops  # noqa: E402
from speech_to_image_translation_without_text_amd._lib import CONV_K1, CONV_K3S1, CONV_K4S2, TCONV_K4S2  # noqa: E402

dev = torch.device("cuda:0")
B = 24
CASES = [
    # name, kind, x shape (NHWC), weight OIHW, wmode, flip
    ("fwd k4s2 64->128 @128 (D256, dominant)", CONV_K4S2, (B, 128, 128, 64), (128, 64, 4, 4), 0, 0),
    ("fwd k4s2 512->1024 @16", CONV_K4S2, (B, 16, 16, 512), (1024, 512, 4, 4), 0, 0),
    ("dgr tconv 256<-512 @16", TCONV_K4S2, (B, 16, 16, 512), (512, 256, 4, 4), 1, 0),
    ("fwd k3s1 64->128 @64", CONV_K3S1, (B, 64, 64, 64), (128, 64, 3, 3), 0, 0),
    ("dgr k3s1 64<-128 @64", CONV_K3S1, (B, 64, 64, 128), (128, 64, 3, 3), 1, 1),
    ("fwd k3s1 32->64 @128", CONV_K3S1, (B, 128, 128, 32), (64, 32, 3, 3), 0, 0),
    ("k1 gemm 4096x2048x1024", CONV_K1, (4, 32, 32, 2048), (1024, 2048), 0, 0),
]
which = sys.argv[1:] or None
g = torch.Generator(device=dev).manual_seed(0)
for name, kind, xs, wshape, wmode, flip in CASES:
    if which and not any(w in name for w in which):
        continue
    x = torch.randn(xs, device=dev, generator=g)
    w = torch.randn(wshape, device=dev, generator=g) / (wshape[1] * (wshape[2] * wshape[3] if len(wshape) == 4 else 1)) ** 0.5
    packed = ops.pack_weight(w, ops.PACK_PLAIN)
    N = wshape[1] if wmode else wshape[0]
    T = {CONV_K1: 1, CONV_K3S1: 9, CONV_K4S2: 16, TCONV_K4S2: 4}[kind]
    Bx, H, W, Cx = xs
    Mout = Bx * H * W * (4 if kind == TCONV_K4S2 else 1) // (4 if kind == CONV_K4S2 else 1)
    flops = 2.0 * Mout * N * T * Cx
    outs = {}
    line = "%-40s" % name
    for planes in (0, 2, 3):
        ops.MATH_PLANES = planes
        fn = lambda: ops.conv_raw(kind, x, None, packed, N, wmode=wmode, flip=flip, wR=packed.shape[1], ldw=packed.shape[2])[0]
        for _ in range(3):
            y = fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            y = fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        outs[planes] = y
        line += "  [%d] %.3f ms %6.1f TF" % (planes, ms, flops / ms / 1e9)
    print(line)
    if kind == CONV_K1:
        ref = (x.view(-1, Cx).double() @ w.double().t()).view(outs[0].shape)
    else:
        ref = outs[0].double()
    scale = float(ref.abs().mean())
    errs = ["%d: max %.2e mean %.2e" % (pl, float((outs[pl].double() - ref).abs().max()) / scale,
                                        float((outs[pl].double() - ref).abs().mean()) / scale) for pl in (0, 2, 3)]
    print("    error / mean|y| vs %s:  %s" % ("fp64" if kind == CONV_K1 else "native fp32", "   ".join(errs)))
