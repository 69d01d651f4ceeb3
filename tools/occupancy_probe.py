import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from speech_to_image_translation_without_text_amd import ops
from speech_to_image_translation_without_text_amd._lib import CONV_K4S2
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
w = torch.randn(128, 64, 4, 4, device=dev, generator=g) / 32
packed = ops.pack_weight(w, ops.PACK_PLAIN)
for B in (8, 16, 24, 48):
    x = torch.randn(B, 128, 128, 64, device=dev, generator=g)
    line = "B=%2d blocks=%4d" % (B, B * 64 * 64 // 128)
    for planes in (0, 3, 1):
        ops.MATH_PLANES = planes
        fn = lambda: ops.conv_raw(CONV_K4S2, x, None, packed, 128, wR=packed.shape[1], ldw=packed.shape[2])[0]
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        line += "  [%d] %.3f ms" % (planes, e0.elapsed_time(e1) / 20)
    print(line)
