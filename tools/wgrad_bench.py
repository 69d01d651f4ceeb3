"""Thin-layer weight-gradient launches in isolation (dev tool, GPU box only); run under rocprofv3 --kernel-trace --stats
to split the time between the GEMM kernel, the slab sum and the OIHW finish."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import _lib, ops  # noqa: E402
from speech_to_image_translation_without_text_amd._lib import CONV_K3S1, CONV_K4S2  # noqa: E402

dev = torch.device("cuda:0")
B = 24
CASES = [
    ("k3 32->64 @128", CONV_K3S1, (B, 128, 128, 32), (B, 128, 128, 64), (64, 32, 3, 3)),
    ("k3 32->32 @128", CONV_K3S1, (B, 128, 128, 32), (B, 128, 128, 32), (32, 32, 3, 3)),
    ("k3 64->128 @64", CONV_K3S1, (B, 64, 64, 64), (B, 64, 64, 128), (128, 64, 3, 3)),
    ("k3 64->64 @64", CONV_K3S1, (B, 64, 64, 64), (B, 64, 64, 64), (64, 64, 3, 3)),
    ("k4 64->128 @128 (D256)", CONV_K4S2, (3 * B, 128, 128, 64), (3 * B, 64, 64, 128), (128, 64, 4, 4)),
    ("k4 128->256 @64 (D256)", CONV_K4S2, (3 * B, 64, 64, 128), (3 * B, 32, 32, 256), (256, 128, 4, 4)),
    ("k4 256->512 @32 (D256)", CONV_K4S2, (3 * B, 32, 32, 256), (3 * B, 16, 16, 512), (512, 256, 4, 4)),
    ("k4 512->1024 @16 (D256)", CONV_K4S2, (3 * B, 16, 16, 512), (3 * B, 8, 8, 1024), (1024, 512, 4, 4)),
    ("k4 1024->2048 @8 (D256)", CONV_K4S2, (3 * B, 8, 8, 1024), (3 * B, 4, 4, 2048), (2048, 1024, 4, 4)),
    ("k3 2048->1024 @4 (D256)", CONV_K3S1, (3 * B, 4, 4, 2048), (3 * B, 4, 4, 1024), (1024, 2048, 3, 3)),
    ("k4 64->128 @128 (G pass)", CONV_K4S2, (B, 128, 128, 64), (B, 64, 64, 128), (128, 64, 4, 4)),
    ("k4 512->1024 @8 (D128)", CONV_K4S2, (3 * B, 8, 8, 512), (3 * B, 4, 4, 1024), (1024, 512, 4, 4)),
]
BMS = [int(v) for v in os.environ.get("WGRAD_BM", "0").split(",")]
which = sys.argv[1:] or None
for name, kind, ashape, gshape, wshape in CASES:
    if which and not any(w in name for w in which):
        continue
    a = torch.randn(ashape, device=dev)
    g = torch.randn(gshape, device=dev)
    out = torch.zeros(wshape, device=dev)
    T = wshape[2] * wshape[3]
    M = gshape[0] * gshape[1] * gshape[2]
    flops = 2.0 * M * wshape[0] * wshape[1] * T
    line = "%-26s" % name
    for bm in BMS:
        with _lib.tuning(wgrad_bm=bm):
            fn = lambda: ops.wgrad_raw(kind, a, None, g, wshape, out=out, accumulate=True)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
        line += " | bm %3d: %7.3f ms  %6.1f TF" % (bm, ms, flops / ms / 1e9)
    print(line, flush=True)
