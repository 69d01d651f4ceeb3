"""Weight gradients on companion streams, eager step (dev tool, GPU box only).  usage: wgrad_stream_probe.py [batch] [bf16]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import model, ops, trainer as T
from speech_to_image_translation_without_text_amd.miscc.config import cfg, cfg_from_file
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ops.ACT_BF16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
cfg_from_file(os.path.join(ROOT, "speech_to_image_translation_without_text_amd", "cfg", "birds_3stages.yml"))
torch.manual_seed(0)
netG = model.G_NET(); netG.apply(T.weights_init)
netsD = [c() for c in (model.D_NET64, model.D_NET128, model.D_NET256)]
[d.apply(T.weights_init) for d in netsD]
netG.to(dev); [d.to(dev) for d in netsD]
tr = T.condGANTrainer(None, None, 256, False); tr.build(netG, netsD)
g = torch.Generator(device=dev).manual_seed(1)
real = [torch.rand(B, 3, 64 << i, 64 << i, device=dev, generator=g) * 2 - 1 for i in range(3)]
wrong = [torch.rand(B, 3, 64 << i, 64 << i, device=dev, generator=g) * 2 - 1 for i in range(3)]
emb = torch.randn(B, 1024, device=dev, generator=g); labels = (torch.arange(B, device=dev) % 3).to(torch.int32)
noise = torch.randn(B, 100, device=dev, generator=g); eps = torch.randn(B, 128, device=dev, generator=g)


def run(n=16):
    for _ in range(4):
        tr.train_step(real, wrong, emb, labels, noise, eps)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(real, wrong, emb, labels, noise, eps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


orig = tr._train_step
for mode in ("off", "all", "off", "g-only", "d-only"):
    if mode == "off":
        tr._train_step = orig
    elif mode == "all":
        def step(*a, **k):
            ops.WGRAD_SIDE_STREAM = True
            try:
                return orig(*a, **k)
            finally:
                ops.WGRAD_SIDE_STREAM = False
        tr._train_step = step
    else:
        og, od = tr._g_backward, tr.train_Dnet

        def wrap(fn):
            def inner(*a, **k):
                ops.WGRAD_SIDE_STREAM = True
                try:
                    return fn(*a, **k)
                finally:
                    ops.WGRAD_SIDE_STREAM = False
            return inner
        tr._train_step = orig
        if mode == "g-only":
            tr._g_backward = wrap(og)
        else:
            tr._g_backward = og
            tr.train_Dnet = wrap(od)
    print("weight gradients on companion streams %-7s: %.2f ms/step" % (mode, run()), flush=True)
