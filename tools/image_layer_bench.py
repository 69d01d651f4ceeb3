"""Timing of the 3-channel image layers at full size (dev tool, GPU box only): GET_IMAGE_G forward / its gradients and the image
gradient of the discriminators' first conv.  usage: image_layer_bench.py [batch] [bf16]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import ops
from speech_to_image_translation_without_text_amd._lib import ACT_LRELU, ACT_TANH
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
bf = len(sys.argv) > 2 and sys.argv[2] == "bf16"
ops.ACT_BF16 = bf
dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for S, ndf in ((256, 64), (128, 64), (64, 64)):
    # first discriminator conv: NHWC4 image -> ndf channels, and its input gradient (the image gradient of the G update)
    x = torch.randn(B, S, S, 4, device=dev).requires_grad_(True)
    w = (torch.randn(ndf, 3, 4, 4, device=dev) * 0.05).requires_grad_(True)
    out = ops.ConvAct.apply(x, w, None, "k4s2", ACT_LRELU, ndf)
    g = torch.randn_like(out)
    fwd = timeit(lambda: ops.ConvAct.apply(x, w, None, "k4s2", ACT_LRELU, ndf))
    w.requires_grad_(False)
    out = ops.ConvAct.apply(x, w, None, "k4s2", ACT_LRELU, ndf)
    bwd = timeit(lambda: torch.autograd.grad(out, x, g, retain_graph=True))
    mb = (x.numel() * 4 + out.numel() * out.element_size()) / 1e6
    print("D first conv %3d px: forward %6.1f us, image gradient (act backward + dgrad) %6.1f us   [%.0f MB each way]" % (S, fwd, bwd, mb))
for S, c in ((256, 16), (128, 32), (64, 64)):
    h = torch.randn(B, S, S, c, device=dev).to(torch.bfloat16 if bf else torch.float32).requires_grad_(True)
    w = (torch.randn(3, c, 3, 3, device=dev) * 0.05).requires_grad_(True)
    out = ops.ConvAct.apply(h, w, None, "k3s1", ACT_TANH, 4)
    g = torch.randn_like(out)
    fwd = timeit(lambda: ops.ConvAct.apply(h, w, None, "k3s1", ACT_TANH, 4))
    bwd = timeit(lambda: torch.autograd.grad(out, (h, w), g, retain_graph=True))
    mb = (h.numel() * h.element_size() + out.numel() * 4) / 1e6
    print("GET_IMAGE_G %3d px from %2d ch: forward %6.1f us, backward (tanh + dgrad + wgrad) %6.1f us   [%.0f MB]" % (S, c, fwd, bwd, mb))
