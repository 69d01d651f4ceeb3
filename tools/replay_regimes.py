"""Step time of the eager step and of the launch-plan replay under different host regimes (dev tool, GPU box only):
sync every k steps / never, main stream = default or a side stream.  usage: replay_regimes.py [batch] [bf16]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import model, ops, trainer as T
from speech_to_image_translation_without_text_amd.miscc.config import cfg, cfg_from_file
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ops.ACT_BF16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
cfg_from_file(os.path.join(ROOT, "speech_to_image_translation_without_text_amd", "cfg", "birds_3stages.yml"))
g = torch.Generator(device=dev).manual_seed(1)
real = [torch.rand(B, 3, 64 << i, 64 << i, device=dev, generator=g) * 2 - 1 for i in range(3)]
wrong = [torch.rand(B, 3, 64 << i, 64 << i, device=dev, generator=g) * 2 - 1 for i in range(3)]
emb = torch.randn(B, 1024, device=dev, generator=g); labels = (torch.arange(B, device=dev) % 3).to(torch.int32)
noise = torch.randn(B, 100, device=dev, generator=g); eps = torch.randn(B, 128, device=dev, generator=g)


def make(graph):
    torch.manual_seed(0)
    netG = model.G_NET(); netG.apply(T.weights_init)
    netsD = [c() for c in (model.D_NET64, model.D_NET128, model.D_NET256)]
    [d.apply(T.weights_init) for d in netsD]
    netG.to(dev); [d.to(dev) for d in netsD]
    tr = T.condGANTrainer(None, None, 256, False); tr.build(netG, netsD)
    if graph:
        tr.enable_graph(warmup=2, executor=graph)
    return tr


def run(tr, n, sync_every, stream=None):
    step = lambda: tr.train_step(real, wrong, emb, labels, noise, eps)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(6): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(n):
            step()
            if sync_every and (k + 1) % sync_every == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


which = sys.argv[3] if len(sys.argv) > 3 else "all"
if which in ("all", "eager"):
    tr = make(None)
    print("eager  main=default sync never: %.2f ms/step" % run(tr, 16, 0), flush=True)
    extra = torch.cuda.Stream()
    with torch.cuda.stream(extra):
        torch.zeros(4, device=dev).add_(1)          # the extra stream has been USED once
    torch.cuda.synchronize()
    print("eager  main=default, an extra used stream exists: %.2f ms/step" % run(tr, 16, 0), flush=True)
    print("eager  main=extra stream: %.2f ms/step" % run(tr, 16, 0, extra), flush=True)
    print("eager  main=default again: %.2f ms/step" % run(tr, 16, 0), flush=True)
    del tr
if which in ("all", "plan"):
    tr = make("plan")
    for sync_every in (1, 0):
        print("plan   main=default sync every %d: %.2f ms/step" % (sync_every, run(tr, 16, sync_every)), flush=True)
    # bounded run-ahead: before enqueueing step k wait for the END of step k - 2 (one step always queued behind the running one)
    step = lambda: tr.train_step(real, wrong, emb, labels, noise, eps)
    for lag in (1, 2):
        evs = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(16):
            if len(evs) >= lag:
                evs[-lag].synchronize()
            step()
            e = torch.cuda.Event(); e.record(); evs.append(e)
        torch.cuda.synchronize()
        print("plan   run-ahead bounded to %d step(s): %.2f ms/step" % (lag, (time.perf_counter() - t0) * 1e3 / 16), flush=True)
