"""96- against 128-row tiles of the fp32 matrix kernel on the launches of the stacked discriminator passes
(dev tool, GPU box only).  Prints ms and TFLOP/s for tile_rows = 128, 96 and the planner's own choice (0)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import ops  # noqa: E402
from speech_to_image_translation_without_text_amd._lib import CONV_K3S1, CONV_K4S2, TCONV_K4S2  # noqa: E402

dev = torch.device("cuda:0")
K4, K3, TC = CONV_K4S2, CONV_K3S1, TCONV_K4S2
CASES = [
    # kind, x shape, N, wmode (1 = input-gradient form)
    (TC, (72, 4, 4, 2048), 1024, 1), (TC, (72, 8, 8, 1024), 512, 1), (K4, (72, 16, 16, 512), 1024, 0),
    (K4, (72, 64, 64, 128), 256, 0), (K4, (72, 32, 32, 256), 512, 0), (TC, (72, 64, 64, 128), 64, 1),
    (TC, (72, 16, 16, 512), 256, 1), (K4, (72, 8, 8, 1024), 2048, 0), (K4, (72, 128, 128, 64), 128, 0),
    (TC, (72, 32, 32, 256), 128, 1), (K3, (72, 4, 4, 2048), 1024, 0), (K3, (72, 4, 4, 1024), 2048, 1),
    (K4, (72, 32, 32, 128), 256, 0), (TC, (72, 8, 8, 512), 256, 1), (K4, (72, 64, 64, 64), 128, 0),
    (TC, (72, 16, 16, 256), 128, 1), (K4, (72, 8, 8, 512), 1024, 0), (K4, (72, 16, 16, 256), 512, 0),
    (K4, (24, 128, 128, 64), 128, 0), (K4, (24, 64, 64, 128), 256, 0), (K4, (24, 8, 8, 1024), 2048, 0),
    (K3, (24, 4, 4, 512), 512, 0), (K3, (24, 4, 4, 1024), 512, 0), (K3, (24, 4, 4, 2048), 1024, 0),
    (K3, (24, 64, 64, 64), 128, 0), (K3, (24, 128, 128, 32), 64, 0),
]
NAME = {K4: "k4s2", K3: "k3s1", TC: "tconv"}
reps = int(os.environ.get("REPS", "10"))
ROWS = tuple(int(v) for v in os.environ.get("ROWS", "128,96,0").split(","))
tot = {r: 0.0 for r in ROWS}
for kind, xs, N, wmode in CASES:
    x = torch.randn(xs, device=dev)
    T = {K3: 9, K4: 16, TC: 16}[kind]
    Cx = xs[3]
    packed = torch.randn((T, N, Cx) if wmode else (T, Cx, (N + 3) & ~3), device=dev) * 0.05
    Tg = {K3: 9, K4: 16, TC: 4}[kind]
    Mout = xs[0] * xs[1] * xs[2] * (4 if kind == TC else 1) // (4 if kind == K4 else 1)
    flops = 2.0 * Mout * N * Tg * Cx
    line = "%-5s x%-22s N%-5d wm%d " % (NAME[kind], list(xs), N, wmode)
    best_ms = {}
    for rows in ROWS + ROWS:      # two alternating passes, the better one counts (the first runs cold)
        ops.TILE_ROWS = rows
        fn = lambda: ops.conv_raw(kind, x, None, packed, N, wmode=wmode, wR=packed.shape[1], ldw=packed.shape[2],
                                  stats=(wmode == 0))
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        best_ms[rows] = min(ms, best_ms.get(rows, 1e9))
    for rows in ROWS:
        ms = best_ms[rows]
        tot[rows] += ms
        line += " | %3d: %6.3f ms %6.1f TF" % (rows, ms, flops / ms / 1e9)
    ops.TILE_ROWS = 0
    print(line, flush=True)
print("sum: " + ", ".join("%s %.2f ms" % ("planner" if r == 0 else "%d rows" % r, tot[r]) for r in ROWS))
