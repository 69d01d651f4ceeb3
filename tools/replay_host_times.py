"""Host time of the launch-plan replay per stream piece (dev tool, GPU box only): how long each s2i_plan_replay call
blocks the Python thread, and the step time.  usage: replay_host_times.py [batch] [bf16]"""
import os, sys, time, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import model, ops, trainer as T, _lib
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
tr.enable_graph(warmup=2)
g = torch.Generator(device=dev).manual_seed(1)
real = [torch.rand(B, 3, 64 << i, 64 << i, device=dev, generator=g) * 2 - 1 for i in range(3)]
wrong = [torch.rand(B, 3, 64 << i, 64 << i, device=dev, generator=g) * 2 - 1 for i in range(3)]
emb = torch.randn(B, 1024, device=dev, generator=g); labels = (torch.arange(B, device=dev) % 3).to(torch.int32)
noise = torch.randn(B, 100, device=dev, generator=g); eps = torch.randn(B, 128, device=dev, generator=g)
step = lambda: tr.train_step(real, wrong, emb, labels, noise, eps)
for _ in range(5): step()
torch.cuda.synchronize()
st = tr._graph
print("launches per piece:", st['plans']['fwd'][1], [p[1] for p in st['plans']['d']], st['plans']['g'][1])
lib = _lib.load()
orig = lib.s2i_plan_replay
times = []
def timed(plan, stream):
    t0 = time.perf_counter(); rc = orig(plan, stream); times.append((time.perf_counter() - t0) * 1e3); return rc
class Shim:
    def __getattr__(self, k): return timed if k == "s2i_plan_replay" else getattr(lib, k)
_lib._lib = Shim()
for it in range(6):
    times.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
    step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("step %d: host %.2f ms, total %.2f ms, pieces (fwd, D256, D128, D64, G) ms: %s" % (
        it, (t1 - t0) * 1e3, (t2 - t0) * 1e3, ["%.2f" % t for t in times]), flush=True)
_lib._lib = lib
# back-to-back steps without a sync (the bench's regime)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print("10 steps back to back: %.2f ms / step" % ((time.perf_counter() - t0) * 100))
