"""In-kernel timeline of the bf16 convolution kernel on D256 conv2 (dev tool, GPU box only): launches the diagnostic
instantiation (libs2i_hip_diag.so, `make -C csrc diag`; knob b16_dbg=32: wave 0 of every block stamps s_memtime at the phase
boundaries) and prints where a block's lifetime goes.  Shares are what to read, not the run time of this build (the stamps
fence overlaps the real kernel has).
usage: python tools/conv16_timeline.py [batch] [v1]"""
import os, sys
OUT = os.environ.setdefault("S2I_B16_TIMELINE", "/tmp/conv_timeline.bin")
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libs2i_hip_diag.so")     # the diagnostic build
from speech_to_image_translation_without_text_amd import ops
from speech_to_image_translation_without_text_amd._lib import CONV_K4S2, PACK_PLAIN
V1 = "v1" in sys.argv[2:]
_lib.check(_lib.load().s2i_set_tuning(b"b16_dbg", 32), "s2i_set_tuning")
_lib.check(_lib.load().s2i_set_tuning(b"b16_v2", 0 if V1 else 1), "s2i_set_tuning")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda:0")
x = torch.randn(3 * B, 128, 128, 64, device=dev).to(torch.bfloat16)
w = torch.randn(128, 64, 4, 4, device=dev) * 0.05
packed = ops.pack_weight(w, PACK_PLAIN)
for _ in range(3):
    ops.conv_any(CONV_K4S2, x, packed, 128, wmode=0, stats=True, out_dtype=torch.bfloat16)
torch.cuda.synchronize()
t = np.fromfile(OUT, dtype=np.uint64).reshape(-1, 64).astype(np.int64)
nb = t.shape[0]
print("blocks", nb)
if not V1:
    # conv_bf16_v2_kernel: 0 start, 1 plan done, 2 prologue loads issued, 3 prologue stored + barrier,
    # 4 + 2 s / 5 + 2 s: stage s before / after its closing barrier, 50 / 51 / 52 epilogue
    life = t[:, 52] - t[:, 0]
    print("block lifetime cycles: median %d  p10 %d  p90 %d" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
    seg = lambda a, b: np.median(t[:, b] - t[:, a])
    print("plan %d   prologue load issue %d   prologue wait + LDS stores + barrier %d" % (seg(0, 1), seg(1, 2), seg(2, 3)))
    prev = 3
    tot_loop = tot_bar = 0
    for s_ in range(8):
        a, b = 4 + 2 * s_, 5 + 2 * s_
        lp, br = seg(prev, a), seg(a, b)
        extra = ""
        if s_ == 3:
            extra = "   (chunk boundary: patch stores + barrier follow)"
        print("stage %d: matrix loop with interleaved loads / stores %5d   barrier %4d%s" % (s_, lp, br, extra))
        tot_loop += lp; tot_bar += br
        prev = b
    print("sum: matrix loops %d   barriers %d" % (tot_loop, tot_bar))
    print("epilogue: transpose + barrier %d   y stores %d   stats %d" % (seg(prev, 50), seg(50, 51), seg(51, 52)))
    sys.exit(0)
life = t[:, 52] - t[:, 0]
print("block lifetime cycles: median %d  p10 %d  p90 %d" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
print("kernel span (max end - min start) cycles: %d" % (t[:, 52].max() - t[:, 0].min()))
seg = lambda a, b: np.median(t[:, b] - t[:, a])
print("prologue (plan + first fetch issue): %d" % seg(0, 1))
names = ["wait loads + ds_write", "barrier 1", "issue next loads + addresses", "matrix loop", "barrier 2"]
tot = np.zeros(5)
for st in range(8):
    sb = 2 + 6 * st
    d = [seg(sb + i, sb + i + 1) for i in range(5)]
    tot += d
    print("stage %d: " % st + "  ".join("%s %5d" % (n, v) for n, v in zip(names, d)))
print("sum over stages: " + "  ".join("%s %6d" % (n, v) for n, v in zip(names, tot)))
print("epilogue: transpose+barrier %d  y stores %d  stats %d" % (seg(2 + 6 * 7 + 5, 50), seg(50, 51), seg(51, 52)))
# residency: blocks per (xcc, se, sh, cu) over time
hw, xcc = t[:, 60], t[:, 61] & 0xf
cu = (xcc << 16) | (hw & 0xff00)
order = np.argsort(t[:, 0])
print("distinct (xcc, cu) ids: %d" % len(np.unique(cu)))
starts = np.sort(t[:, 0] - t[:, 0].min())
print("block start times (cycles) at ranks 0, 512, 1024, 2048, 4096: ", [int(starts[min(i, nb - 1)]) for i in (0, 512, 1024, 2048, 4096)])
rt = t[:, 62]
print("s_memrealtime span (100 MHz ticks): %d -> %.1f us; shader clock ~ %.2f GHz" %
      (rt.max() - rt.min(), (rt.max() - rt.min()) / 100.0, (t[:, 52].max() - t[:, 52].min()) / max(1, (rt.max() - rt.min())) * 0.1))
