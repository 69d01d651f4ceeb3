"""Generate tests/golden/*.npz from the REFERENCE itself (runs only in the build container).

The reference's StackGAN_v2/model.py and trainer.py are imported from /root/reference on CPU with
in-memory stand-ins for the three absent third-party names that are off the hot path
(easydict.EasyDict, torchvision's Inception3 base class / utils, tensorboardX.SummaryWriter;
SURVEY.md §8c).  Each case seeds torch, builds the reference networks, applies its weights_init,
runs its own forward / train_Dnet / train_Gnet, and stores inputs' seeds, expected outputs and
parameter checksums.  The CPU oracle (oracle/stackgan_oracle.py) is asserted against the reference
here, at generation time, on the same weights.

Weights are NOT stored: speech_to_image_translation_without_text_amd.model builds the same
torch.nn parameter containers in the same order, so `torch.manual_seed(seed)` + construction +
weights_init reproduces them bit for bit; the checksums stored here prove it in the tests.

Usage:  python tests/golden/make_golden.py            (writes next to this file)
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/StackGAN_v2'

CASES = {
    # reduced-width three-stage nets: every block type, full 64/128/256 geometry, small channels
    'small3': dict(branch=3, gf=16, df=8, ef=16, z=12, t=32, B=4, seed=0, data_seed=1, step=True),
    # BASELINE config 1: branch_num=1 at full width, batch 4
    'full1': dict(branch=1, gf=64, df=64, ef=128, z=100, t=1024, B=4, seed=0, data_seed=1, step=True),
    # full-width three-stage forward (BASELINE config 2 shapes) at batch 2
    'full3_fwd': dict(branch=3, gf=64, df=64, ef=128, z=100, t=1024, B=2, seed=0, data_seed=1, step=False),
}


def import_reference():
    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)
        __setattr__ = __setitem__

    ed = types.ModuleType('easydict'); ed.EasyDict = EasyDict
    tv = types.ModuleType('torchvision'); tvm = types.ModuleType('torchvision.models')
    tvu = types.ModuleType('torchvision.utils'); tvt = types.ModuleType('torchvision.transforms')

    class Inception3(nn.Module):
        pass
    tvm.Inception3 = Inception3
    tv.models, tv.utils, tv.transforms = tvm, tvu, tvt
    tbx = types.ModuleType('tensorboardX')

    class SummaryWriter(object):
        def __init__(self, *a, **k): pass
        def __getattr__(self, name): return lambda *a, **k: None
    tbx.SummaryWriter = SummaryWriter
    for name, mod in (('easydict', ed), ('torchvision', tv), ('torchvision.models', tvm), ('torchvision.utils', tvu),
                      ('torchvision.transforms', tvt), ('tensorboardX', tbx)):
        sys.modules.setdefault(name, mod)
    sys.path.insert(0, REF)
    import miscc.config as rcfg
    import model as rmodel
    import trainer as rtrainer
    return rcfg.cfg, rmodel, rtrainer


def set_cfg(cfg, c):
    cfg.CUDA = False
    cfg.TREE.BRANCH_NUM = c['branch']
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM = c['gf'], c['df']
    cfg.GAN.EMBEDDING_DIM, cfg.GAN.Z_DIM, cfg.TEXT.DIMENSION = c['ef'], c['z'], c['t']
    cfg.GAN.R_NUM, cfg.GAN.B_CONDITION = 2, True
    cfg.TRAIN.BATCH_SIZE = c['B']
    cfg.TRAIN.COEFF.UNCOND_LOSS, cfg.TRAIN.COEFF.CAL_LOSS = 1.0, 50.0
    cfg.TRAIN.COEFF.KL, cfg.TRAIN.COEFF.COLOR_LOSS = 2.0, 0.0
    cfg.TRAIN.DISCRIMINATOR_LR = cfg.TRAIN.GENERATOR_LR = 2e-4
    cfg.TRAIN.LOG_INTERVAL = 100


def make_batch(c):
    """Synthetic batch of SURVEY.md §8d: N(0,1) embeddings/noise/eps, U(-1,1) images, labels i % 3."""
    g = torch.Generator().manual_seed(c['data_seed'])
    B = c['B']
    batch = dict(emb=torch.randn(B, c['t'], generator=g), noise=torch.randn(B, c['z'], generator=g),
                 eps=torch.randn(B, c['ef'], generator=g), real=[], wrong=[],
                 labels=[i % 3 for i in range(B)])
    for i in range(c['branch']):
        s = 64 << i
        batch['real'].append(torch.rand(B, 3, s, s, generator=g) * 2 - 1)
        batch['wrong'].append(torch.rand(B, 3, s, s, generator=g) * 2 - 1)
    return batch


def build_reference_nets(rmodel, rtrainer, c):
    torch.manual_seed(c['seed'])
    netG = rmodel.G_NET(); netG.apply(rtrainer.weights_init)
    netsD = []
    for i, cls in enumerate((rmodel.D_NET64, rmodel.D_NET128, rmodel.D_NET256)[:c['branch']]):
        d = cls(); d.apply(rtrainer.weights_init); netsD.append(d)
    return netG, netsD


def checksum(sd):
    """Order-sensitive fingerprint of a state_dict: per-tensor (sum, sum of squares, first, last)."""
    rows = []
    for k, v in sd.items():
        f = v.detach().double().reshape(-1)
        rows.append([float(f.sum()), float((f * f).sum()), float(f[0]), float(f[-1])])
    return np.asarray(rows, dtype=np.float64)


def sample(t, n=4096):
    """Deterministic strided sample of a tensor (keeps fixtures small for big tensors)."""
    f = t.detach().reshape(-1)
    if f.numel() <= n:
        return f.numpy().copy()
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return f[idx].numpy().copy()


def run_case(name, c, cfg, rmodel, rtrainer):
    sys.path.insert(0, ROOT)
    from oracle import stackgan_oracle as orc
    set_cfg(cfg, c)
    netG, netsD = build_reference_nets(rmodel, rtrainer, c)
    batch = make_batch(c)
    out = {'cfg': np.asarray([c[k] for k in ('branch', 'gf', 'df', 'ef', 'z', 't', 'B', 'seed', 'data_seed')])}
    out['g_keys'] = np.asarray(list(netG.state_dict().keys()))
    out['g_checksum'] = checksum(netG.state_dict())
    for i, d in enumerate(netsD):
        out['d%d_keys' % i] = np.asarray(list(d.state_dict().keys()))
        out['d%d_checksum' % i] = checksum(d.state_dict())
    dims = orc.Dims(c['branch'], c['gf'], c['df'], c['ef'], c['z'], c['t'], 2)
    ostate = orc.TrainState({k: v.clone() for k, v in netG.state_dict().items()},
                            [{k: v.clone() for k, v in d.state_dict().items()} for d in netsD])

    # the reference draws eps from the global RNG inside CA_NET (model.py:193): pin it by seeding
    def seed_eps():
        torch.manual_seed(4242)
    seed_eps()
    eps = torch.FloatTensor(c['B'], c['ef']).normal_()
    batch['eps'] = eps
    out['eps'] = eps.numpy()

    if not c['step']:
        seed_eps()
        fakes, mu, logvar = netG(batch['noise'], batch['emb'])
        ofakes, omu, olv = orc.g_forward(dict(ostate.g), batch['noise'], batch['emb'], eps, dims)
        for i, f in enumerate(fakes):
            assert torch.allclose(f, ofakes[i], rtol=1e-4, atol=1e-5), 'oracle != reference (img %d)' % i
            out['fake%d_sample' % i] = sample(f, 16384)
            out['fake%d_stats' % i] = np.asarray([float(f.double().mean()), float(f.double().std())])
        out['mu'], out['logvar'] = mu.detach().numpy(), logvar.detach().numpy()
        for i, d in enumerate(netsD):
            logits, feat = d(fakes[i].detach(), mu.detach())
            ol, ofeat = orc.d_forward(dict(ostate.ds[i]), 64 << i, fakes[i].detach(), mu.detach())
            assert torch.allclose(logits[0], ol[0], rtol=1e-4, atol=1e-5)
            out['d%d_cond' % i], out['d%d_uncond' % i] = logits[0].detach().numpy(), logits[1].detach().numpy()
            out['d%d_feat_sample' % i] = sample(feat)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
        return

    # ---- one full iteration through the reference's own train_Dnet / train_Gnet -------------------
    T = rtrainer.condGANTrainer
    tr = object.__new__(T)
    tr.netG, tr.netsD, tr.num_Ds = netG, netsD, len(netsD)
    tr.optimizerG, tr.optimizersD = rtrainer.define_optimizers(netG, netsD)
    tr.criterion = nn.BCELoss()
    tr.real_labels = torch.ones(c['B']); tr.fake_labels = torch.zeros(c['B'])
    tr.summary_writer = sys.modules['tensorboardX'].SummaryWriter()
    tr.real_imgs = [t.clone().requires_grad_() for t in batch['real']]
    tr.wrong_imgs = [t.clone().requires_grad_() for t in batch['wrong']]
    tr.txt_embedding = batch['emb'].clone().requires_grad_()
    tr.class_labels = batch['labels']
    avg_param_G = rtrainer.copy_G_params(netG)
    seed_eps()
    tr.fake_imgs, tr.mu, tr.logvar = netG(batch['noise'].clone().requires_grad_(True), tr.txt_embedding)
    for i, f in enumerate(tr.fake_imgs):
        out['fake%d_sample' % i] = sample(f, 16384)
        if f.numel() <= 60000:
            out['fake%d' % i] = f.detach().numpy()
    out['mu'], out['logvar'] = tr.mu.detach().numpy(), tr.logvar.detach().numpy()
    errD = [float(tr.train_Dnet(i, 1)) for i in range(tr.num_Ds)]
    kl, errG_total = tr.train_Gnet(1)
    for p_, avg_p in zip(netG.parameters(), avg_param_G):
        avg_p.mul_(0.999).add_(p_.data, alpha=0.001)
    out['errD'] = np.asarray(errD)
    out['errG_total'], out['kl'] = np.asarray(float(errG_total)), np.asarray(float(kl))
    out['grad_emb'] = tr.txt_embedding.grad.numpy()
    out['g_after_checksum'] = checksum(netG.state_dict())
    for i, d in enumerate(netsD):
        out['d%d_after_checksum' % i] = checksum(d.state_dict())
    # a few directly comparable tensors after the update
    gsd = netG.state_dict()
    for k in ('ca_net.fc.weight', 'h_net1.upsample4.1.weight', 'img_net1.img.0.weight', 'h_net1.fc.1.running_var'):
        out['g_after/' + k] = sample(gsd[k])
    for k, g_ in zip(dict(netG.named_parameters()).keys(), [p_.grad for p_ in netG.parameters()]):
        if k in ('ca_net.fc.weight', 'h_net1.fc.0.weight', 'h_net1.upsample1.1.weight', 'img_net1.img.0.weight') or \
                k.endswith('jointConv.0.weight') or k.endswith('upsample.1.weight') or k.endswith('block.3.weight'):
            out['g_grad/' + k] = sample(g_)
    for i, d in enumerate(netsD):
        dsd = d.state_dict()
        for k in ('img_code_s16.0.weight', 'img_code_s16.8.weight', 'jointConv.0.weight', 'logits.0.bias',
                  'img_code_s16.9.running_mean'):
            out['d%d_after/%s' % (i, k)] = sample(dsd[k])
    out['avg_g/ca_net.fc.weight'] = sample(avg_param_G[0])

    # oracle against the reference, same weights, same batch
    o = orc.train_step(ostate, batch, dims)
    assert np.allclose(o['errD'], errD, rtol=2e-4, atol=1e-5), (o['errD'], errD)
    assert abs(o['errG_total'] - float(errG_total)) <= 2e-4 * abs(float(errG_total)) + 1e-5, (o['errG_total'], errG_total)
    assert torch.allclose(o['grad_emb'], tr.txt_embedding.grad, rtol=1e-3, atol=1e-6)
    for k, v in gsd.items():
        assert torch.allclose(ostate.g[k].float(), v.float(), rtol=1e-3, atol=2e-5), 'oracle G param %s' % k
    for i, d in enumerate(netsD):
        for k, v in d.state_dict().items():
            assert torch.allclose(ostate.ds[i][k].float(), v.float(), rtol=1e-3, atol=2e-5), 'oracle D%d %s' % (i, k)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


def main():
    torch.set_num_threads(8)
    cfg, rmodel, rtrainer = import_reference()
    for name, c in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        run_case(name, c, cfg, rmodel, rtrainer)
        print('wrote', name, os.path.getsize(os.path.join(HERE, name + '.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
