"""Soak: many train steps in one process, watching for non-finite losses and allocator growth (dev tool, GPU box only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import model, ops, trainer as T  # noqa: E402
from speech_to_image_translation_without_text_amd.miscc.config import cfg, cfg_from_file  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ops.MATH_PLANES = int(os.environ.get("S2I_MATH_PLANES", "0"))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 24      # usage: soak.py [steps] [batch]; S2I_ACT_BF16=1 for the bf16 mode
dev = torch.device("cuda:0")
cfg_from_file(os.path.join(ROOT, "speech_to_image_translation_without_text_amd", "cfg", "birds_3stages.yml"))
cfg.TRAIN.BATCH_SIZE = B
torch.manual_seed(0)
netG = model.G_NET(); netG.apply(T.weights_init)
netsD = [c() for c in (model.D_NET64, model.D_NET128, model.D_NET256)]
for d in netsD:
    d.apply(T.weights_init)
netG.to(dev); [d.to(dev) for d in netsD]
tr = T.condGANTrainer(None, None, 256, False); tr.build(netG, netsD)
g = torch.Generator(device=dev).manual_seed(1)
real = [torch.rand(B, 3, 64 << i, 64 << i, device=dev, generator=g) * 2 - 1 for i in range(3)]
wrong = [torch.rand(B, 3, 64 << i, 64 << i, device=dev, generator=g) * 2 - 1 for i in range(3)]
emb = torch.randn(B, 1024, device=dev, generator=g)
labels = (torch.arange(B, device=dev) % 3).to(torch.int32)
noise = torch.empty(B, 100, device=dev); eps = torch.empty(B, 128, device=dev)
marks = {}
for it in range(steps):
    noise.normal_(generator=g); eps.normal_(generator=g)
    out = tr.train_step(real, wrong, emb.detach().requires_grad_(True), labels, noise, eps)
    if it in (20, steps // 2, steps - 1):
        torch.cuda.synchronize()
        vals = [float(o) for o in out]
        assert all(v == v and abs(v) < 1e6 for v in vals), vals
        marks[it] = (torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, vals)
        print("step %4d: allocated %d MiB reserved %d MiB  errD %.4f errG %.4f kl %.4f" % ((it,) + marks[it][:2] + tuple(vals)))
a = [m[0] for m in marks.values()]
assert max(a) - min(a) < 64, "allocated memory drifts: %s" % a
print("soak ok")
