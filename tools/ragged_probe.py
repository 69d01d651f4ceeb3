"""Is the HIP train step at a ragged batch (5, 23) or a plain one (8) independent of what the allocator hands back?  Runs the
step of tests/test_parity_gpu.py::test_ragged_batch_full_train_step twice, the second time after filling and freeing 4 GiB
with NaN / 1e30 / 0, and compares every gradient and updated weight bitwise (dev tool, GPU box only)."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from helpers import *  # noqa
import test_parity_gpu as tp
from speech_to_image_translation_without_text_amd import ops, trainer as T, _lib
_lib.load()
gpu = torch.device("cuda:0")
def run(B, poison):
    case = dict(tp.CASES['small3'], B=B)
    netG, netsD = tp.build_nets(case)
    batch = tp.make_batch(case)
    netG.to(gpu); [d.to(gpu) for d in netsD]
    if poison is not None:
        junk = [torch.full((64 << 20,), poison, device=gpu) for _ in range(8)]
        del junk
    tr = T.condGANTrainer(None, None, 256, False); tr.build(netG, netsD); tr.flatG.lr = 0.0
    b = tp.to_dev(batch, gpu)
    emb = b['emb'].clone().requires_grad_(True)
    with ops.param_grad_mode(True):
        tr.real_imgs, tr.wrong_imgs, tr.class_labels = b['real'], b['wrong'], batch['labels']
        tr.fake_imgs, tr.mu, tr.logvar = netG(b['noise'], emb, b['eps'])
        errD = sum(tr.train_Dnet(i, 0) for i in range(3))
        kl, errG = tr.train_Gnet(0)
    torch.cuda.synchronize()
    out = {"emb": emb.grad.clone(), "errD": torch.tensor(float(errD)), "errG": torch.tensor(float(errG))}
    for k, p in netG.named_parameters():
        out["g/" + k] = p.grad.clone()
    for i, d in enumerate(netsD):
        for k, p in d.named_parameters():
            out["d%d/%s" % (i, k)] = p.detach().clone()
    return out
for B in (5, 23, 8):
    a = run(B, None)
    for poison in (float("nan"), 1e30, 0.0):
        c = run(B, poison)
        bad = [(k, float((a[k].float() - c[k].float()).abs().max())) for k in a if not torch.equal(a[k], c[k])]
        print("B=%d poison=%s: %d of %d tensors differ" % (B, poison, len(bad), len(a)), bad[:6])
