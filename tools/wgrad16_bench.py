"""bf16 weight-gradient launches of the config-4 step in isolation, 256- against 128-row tiles (dev tool, GPU box only).
Each line: layer, ms and TFLOP/s for wgrad16_bm = 128, 256 (forced where 256 divides K) and 0 (the planner's cost model), and
the largest relative difference of the forced results."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import _lib, ops  # noqa: E402
from speech_to_image_translation_without_text_amd._lib import CONV_K3S1, CONV_K4S2  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
S = 3 * B
CASES = [
    ("D256 conv2 k4 64->128 @128", CONV_K4S2, (S, 128, 128, 64), (S, 64, 64, 128), (128, 64, 4, 4)),
    ("D256 conv3 k4 128->256 @64", CONV_K4S2, (S, 64, 64, 128), (S, 32, 32, 256), (256, 128, 4, 4)),
    ("D256 conv4 k4 256->512 @32", CONV_K4S2, (S, 32, 32, 256), (S, 16, 16, 512), (512, 256, 4, 4)),
    ("D256 s32 k4 512->1024 @16", CONV_K4S2, (S, 16, 16, 512), (S, 8, 8, 1024), (1024, 512, 4, 4)),
    ("D256 s64 k4 1024->2048 @8", CONV_K4S2, (S, 8, 8, 1024), (S, 4, 4, 2048), (2048, 1024, 4, 4)),
    ("D256 s64_1 k3 2048->1024 @4", CONV_K3S1, (S, 4, 4, 2048), (S, 4, 4, 1024), (1024, 2048, 3, 3)),
    ("D128 conv2 k4 64->128 @64", CONV_K4S2, (S, 64, 64, 64), (S, 32, 32, 128), (128, 64, 4, 4)),
    ("D128 s32 k4 512->1024 @8", CONV_K4S2, (S, 8, 8, 512), (S, 4, 4, 1024), (1024, 512, 4, 4)),
    ("G up k3 128->256 @32 (B)", CONV_K3S1, (B, 32, 32, 128), (B, 32, 32, 256), (256, 128, 3, 3)),
    ("G res k3 64->128 @128 (B)", CONV_K3S1, (B, 128, 128, 64), (B, 128, 128, 128), (128, 64, 3, 3)),
    ("G joint k3 192->128 @64 (B)", CONV_K3S1, (B, 64, 64, 192), (B, 64, 64, 128), (128, 192, 3, 3)),
]
which = sys.argv[2:] or None
tot = {128: 0.0, 256: 0.0, 0: 0.0, 512: 0.0}
for name, kind, ashape, gshape, wshape in CASES:
    if which and not any(w in name for w in which):
        continue
    a = torch.randn(ashape, device=dev).to(torch.bfloat16)
    g = torch.randn(gshape, device=dev).to(torch.bfloat16)
    res = {}
    line = "%-30s" % name
    T = wshape[2] * wshape[3]
    M = gshape[0] * gshape[1] * gshape[2]
    flops = 2.0 * M * wshape[0] * wshape[1] * T
    for bm in (128, 256, 0) + ((512,) if os.environ.get('TRY512') else ()):
        with _lib.tuning(wgrad16_bm=bm):
            out = torch.zeros(wshape, device=dev)
            fn = lambda: ops.wgrad_any(kind, a, g, wshape, out=out)
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
            res[bm] = out.clone()
        tot[bm] += ms
        line += " | %3d: %6.3f ms %6.1f TF" % (bm, ms, flops / ms / 1e9)
    rel = float((res[128] - res[256]).abs().max() / res[128].abs().max())
    print(line + " | max rel diff %.1e" % rel, flush=True)
print("sum: 128-row tiles %.3f ms, 256-row tiles %.3f ms, planner %.3f ms" % (tot[128], tot[256], tot[0]))
