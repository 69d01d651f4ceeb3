"""Where the host spends its time while enqueuing one train step (cProfile, dev tool, GPU box only).
usage: python tools/host_profile.py [batch] [bf16]"""
import cProfile, os, pstats, sys, io
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
step = lambda: tr.train_step(real, wrong, emb.detach().requires_grad_(True), labels, noise, eps)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
