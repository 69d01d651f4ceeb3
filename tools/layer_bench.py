"""Per-launch timing of every implicit-GEMM launch of one train step (dev tool, GPU box only).

Records the (descriptor, count) of every s2i_conv_forward / s2i_conv_wgrad call during one real
cfg/birds_3stages.yml step, then times each unique descriptor in isolation with HIP events.
"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import model, ops, trainer as T  # noqa: E402
from speech_to_image_translation_without_text_amd.miscc.config import cfg, cfg_from_file  # noqa: E402

KIND = {0: "k1", 1: "k3s1", 2: "k4s2", 3: "tconv"}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
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
    noise = torch.randn(B, 100, device=dev, generator=g); eps = torch.randn(B, 128, device=dev, generator=g)
    step = lambda: tr.train_step(real, wrong, emb.detach().requires_grad_(True), labels, noise, eps)
    step(); torch.cuda.synchronize()

    calls = collections.OrderedDict()
    orig_conv, orig_wgrad = ops.conv_raw, ops.wgrad_raw

    def rec_conv(kind, x, cvec, packed, N, **kw):
        key = ("fwd", kind, tuple(x.shape), 0 if cvec is None else cvec.shape[1], N, kw.get("wmode", 0), kw.get("flip", 0),
               kw["wR"], kw["ldw"], kw.get("act", 0), bool(kw.get("stats", False)), tuple(packed.shape))
        calls[key] = calls.get(key, 0) + 1
        return orig_conv(kind, x, cvec, packed, N, **kw)

    def rec_wgrad(kind, a, cvec, gten, grad_shape, **kw):
        key = ("wgrad", kind, tuple(a.shape), 0 if cvec is None else cvec.shape[1], tuple(gten.shape), tuple(grad_shape),
               kw.get("swap", 0), kw.get("fold", 0))
        calls[key] = calls.get(key, 0) + 1
        return orig_wgrad(kind, a, cvec, gten, grad_shape, **kw)

    ops.conv_raw, ops.wgrad_raw = rec_conv, rec_wgrad
    step(); torch.cuda.synchronize()
    ops.conv_raw, ops.wgrad_raw = orig_conv, orig_wgrad

    rows = []
    for key, cnt in calls.items():
        if key[0] == "fwd":
            _, kind, xs, Cc, N, wmode, flip, wR, ldw, act, stats, ps = key
            x = torch.randn(xs, device=dev); cv = torch.randn(xs[0], Cc, device=dev) if Cc else None
            packed = torch.randn(ps, device=dev) * 0.05
            fn = lambda: orig_conv(kind, x, cv, packed, N, wmode=wmode, flip=flip, wR=wR, ldw=ldw, act=act, stats=stats)
            Bx, H, W, Cx = xs
            T_ = {0: 1, 1: 9, 2: 16, 3: 4}[kind]
            Mout = Bx * H * W * (4 if kind == 3 else 1) // (4 if kind == 2 else 1)
            flops = 2.0 * Mout * N * T_ * (Cx + Cc)
            desc = "%s %-5s x%s Cc%d N%d wm%d" % ("fwd", KIND[kind], list(xs), Cc, N, wmode)
        else:
            _, kind, as_, Cc, gs, gshape, swap, fold = key
            a = torch.randn(as_, device=dev); cv = torch.randn(as_[0], Cc, device=dev) if Cc else None
            gt = torch.randn(gs, device=dev)
            fn = lambda: orig_wgrad(kind, a, cv, gt, gshape, swap=swap, fold=fold)
            Bx, H, W, Ca = as_
            T_ = {0: 1, 1: 9, 2: 16}[kind]
            M = Bx * H * W // (4 if kind == 2 else 1)
            flops = 2.0 * M * gs[-1] * T_ * (Ca + Cc)
            desc = "%s %-5s a%s Cc%d g%s" % ("wgr", KIND[kind], list(as_), Cc, list(gs))
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        rows.append((ms * cnt, ms, cnt, flops, desc))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("total igemm-path time per step: %.2f ms over %d unique launches" % (tot, len(rows)))
    for t, ms, cnt, flops, desc in rows[:70]:
        print("%6.2f ms (%4.1f%%)  %3dx %7.3f ms  %6.1f TF  %s" % (t, 100 * t / tot, cnt, ms, flops / ms / 1e9, desc))


if __name__ == "__main__":
    main()
