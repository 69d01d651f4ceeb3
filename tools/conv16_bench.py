"""Per-layer timing of the bf16 convolution kernels on the shapes of BASELINE config 4 (dev tool, GPU box only).
usage: python tools/conv16_bench.py [batch]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import ops
from speech_to_image_translation_without_text_amd._lib import CONV_K3S1, CONV_K4S2, TCONV_K4S2, PACK_PLAIN, PACK_UPFOLD

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda:0")
# (name, kind, images, H, Cin, Cout, wmode)
LAYERS = [
    ("D256 conv2 fwd", CONV_K4S2, 3 * B, 128, 64, 128, 0), ("D256 conv3 fwd", CONV_K4S2, 3 * B, 64, 128, 256, 0),
    ("D256 conv4 fwd", CONV_K4S2, 3 * B, 32, 256, 512, 0), ("D256 s32 fwd", CONV_K4S2, 3 * B, 16, 512, 1024, 0),
    ("D256 s64 fwd", CONV_K4S2, 3 * B, 8, 1024, 2048, 0), ("D256 s64_1 fwd", CONV_K3S1, 3 * B, 4, 2048, 1024, 0),
    ("D256 conv3 dgrad", TCONV_K4S2, 3 * B, 32, 256, 128, 1), ("D256 conv2 dgrad", TCONV_K4S2, 3 * B, 64, 128, 64, 1),
    ("D256 conv4 dgrad", TCONV_K4S2, 3 * B, 16, 512, 256, 1),
    ("D256 s32 dgrad", TCONV_K4S2, 3 * B, 8, 1024, 512, 1), ("D256 s64 dgrad", TCONV_K4S2, 3 * B, 4, 2048, 1024, 1),
    ("D256 s64_1 dgrad", CONV_K3S1, 3 * B, 4, 1024, 2048, 1), ("D256 s64 fwd (G pass)", CONV_K4S2, B, 8, 1024, 2048, 0),
    ("D256 s32 fwd (G pass)", CONV_K4S2, B, 16, 512, 1024, 0), ("D128 s32 fwd", CONV_K4S2, 3 * B, 8, 512, 1024, 0),
    ("G h2 res conv fwd", CONV_K3S1, B, 64, 64, 128, 0), ("G h2 res conv2 fwd", CONV_K3S1, B, 64, 64, 64, 0),
    ("G h3 res conv fwd", CONV_K3S1, B, 128, 32, 64, 0), ("G h3 res conv2 fwd", CONV_K3S1, B, 128, 32, 32, 0),
    ("G h3 up fwd", TCONV_K4S2, B, 128, 32, 32, 0), ("G h2 up fwd", TCONV_K4S2, B, 64, 64, 64, 0),
    ("G up4 fwd", TCONV_K4S2, B, 32, 128, 128, 0), ("G up3 fwd", TCONV_K4S2, B, 16, 256, 256, 0),
    ("G h3 up dgrad", CONV_K4S2, B, 256, 32, 32, 1), ("G h2 up dgrad", CONV_K4S2, B, 128, 64, 64, 1),
]
print("batch", B)
for name, kind, n, H, Cin, Cout, wmode in LAYERS:
    x = torch.randn(n, H, H, Cin, device=dev).to(torch.bfloat16)
    T = {CONV_K3S1: 9, CONV_K4S2: 16, TCONV_K4S2: 16}[kind]
    kk = 3 if kind == CONV_K3S1 else 4
    if wmode == 0:
        w = torch.randn(Cout, Cin, kk, kk, device=dev) * 0.05
        if kind == TCONV_K4S2:
            w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
            packed = ops.pack_weight(w, PACK_UPFOLD)
        else:
            packed = ops.pack_weight(w, PACK_PLAIN)
    else:
        w = torch.randn(Cin, Cout, kk, kk, device=dev) * 0.05   # forward layer Cout -> Cin channels; dy has Cin channels here
        packed = ops.pack_weight(w, PACK_PLAIN)
    fn = lambda: ops.conv_any(kind, x, packed, Cout, wmode=wmode, stats=(wmode == 0), out_dtype=torch.bfloat16)
    for _ in range(3):
        y = fn()[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    taps = {CONV_K3S1: 9, CONV_K4S2: 16, TCONV_K4S2: 4}[kind]
    flops = 2.0 * y.numel() * taps * Cin
    byts = x.numel() * 2 + y.numel() * 2 + T * Cin * Cout * 2
    print("%-20s %8.1f us  %7.1f TF  %6.2f TB/s (algorithmic %.0f MB)" % (name, ms * 1e3, flops / ms / 1e9, byts / ms / 1e9, byts / 1e6))
