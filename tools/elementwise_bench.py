"""HBM rate of the BatchNorm / activation passes on config 4's largest tensors (dev tool, GPU box only).
usage: python tools/elementwise_bench.py [batch] [f32]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import _lib, ops
from speech_to_image_translation_without_text_amd._lib import ACT_GLU, ACT_LRELU, DT_BF16, DT_F32, ptr, stream, check

lib = ops._lib_ready()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
f32 = len(sys.argv) > 2 and sys.argv[2] == "f32"
dt, tdt, es = (DT_F32, torch.float32, 4) if f32 else (DT_BF16, torch.bfloat16, 2)
dev = torch.device("cuda:0")
# (name, rows M, channels C of the raw conv output y, activation, BatchNorm groups)
CASES = [("D256 conv2 (3B x 64x64, 128 ch, lrelu)", 3 * B * 64 * 64, 128, ACT_LRELU, 3),
         ("D256 conv3 (3B x 32x32, 256 ch, lrelu)", 3 * B * 32 * 32, 256, ACT_LRELU, 3),
         ("G h3 up (B x 256x256, 32 ch, glu)", B * 256 * 256, 32, ACT_GLU, 1),
         ("G h2 up (B x 128x128, 64 ch, glu)", B * 128 * 128, 64, ACT_GLU, 1),
]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, M, C, act, groups in CASES:
    Co = C // 2 if act == ACT_GLU else C
    y = torch.randn(M, C, device=dev).to(tdt)
    out = torch.empty(M, Co, device=dev, dtype=tdt)
    dout = torch.randn(M, Co, device=dev).to(tdt)
    dy = torch.empty(M, C, device=dev, dtype=tdt)
    coef = torch.randn(groups, 4, C, device=dev).abs() + 0.5
    nparts = ops._num_parts(M // groups) * groups
    part = torch.empty(2 * nparts * C, device=dev)
    red2 = torch.randn(groups, 2, C, device=dev)
    t_fwd = timeit(lambda: check(lib.s2i_bn_act_forward_dt(dt, ptr(y), M, groups, C, ptr(coef), act, None, ptr(out), stream()), "fwd"))
    t_red = timeit(lambda: check(lib.s2i_bn_act_bwd_reduce_dt(dt, ptr(y), ptr(dout), Co, M, groups, C, ptr(coef), act, ptr(part), nparts, stream()), "red"))
    t_app = timeit(lambda: check(lib.s2i_bn_act_bwd_apply_dt(dt, ptr(y), ptr(dout), Co, M, groups, C, ptr(coef), ptr(red2), act, ptr(dy), stream()), "app"))
    b_fwd = M * (C + Co) * es
    b_red = M * (C + Co) * es
    b_app = M * (2 * C + Co) * es
    print("%-48s fwd %6.1f us %5.2f TB/s | bwd reduce %6.1f us %5.2f TB/s | bwd apply %6.1f us %5.2f TB/s" %
          (name, t_fwd * 1e3, b_fwd / t_fwd / 1e9, t_red * 1e3, b_red / t_red / 1e9, t_app * 1e3, b_app / t_app / 1e9))
