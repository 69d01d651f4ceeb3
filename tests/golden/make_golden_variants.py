"""Generate tests/golden/variants.npz from the REFERENCE itself (build container only): the configuration
branches of the hot path that the main fixtures (make_golden.py) leave dormant.

  color     cfg.TRAIN.COEFF.COLOR_LOSS = 1.0: one full iteration through the reference's own train_Dnet / train_Gnet
            with the colour-consistency terms of trainer.py:455-478 (compute_mean_covariance :34-51)
  nouncond  cfg.TRAIN.COEFF.UNCOND_LOSS = 0: errD = real + 0.5 * (wrong + fake) (trainer.py:411-412), G loss without
            the unconditional term (:439-443)
  nocond    cfg.GAN.B_CONDITION = False (model.py:308, 332-336, 418, 430-445): no ca_net / jointConv / second head.  The
            reference's trainer cannot run this setting (train_Dnet calls mu.detach() on None), so the fixture is
            model-level: G and D forwards and the gradient of sum(logits) + <x_immediate, r> w.r.t. z and parameters.

Same import stand-ins and seeding as make_golden.py; the CPU oracle is asserted against the reference here.
Usage:  python tests/golden/make_golden_variants.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

CASE = dict(mg.CASES['small3'])


def _one_iteration(cfg, rmodel, rtrainer, c, tag, out):
    from oracle import stackgan_oracle as orc
    netG, netsD = mg.build_reference_nets(rmodel, rtrainer, c)
    batch = mg.make_batch(c)
    dims = orc.Dims(c['branch'], c['gf'], c['df'], c['ef'], c['z'], c['t'], 2)
    ostate = orc.TrainState({k: v.clone() for k, v in netG.state_dict().items()},
                            [{k: v.clone() for k, v in d.state_dict().items()} for d in netsD])
    torch.manual_seed(4242)
    eps = torch.FloatTensor(c['B'], c['ef']).normal_()
    batch['eps'] = eps
    out[tag + '/eps'] = eps.numpy()
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
    torch.manual_seed(4242)
    tr.fake_imgs, tr.mu, tr.logvar = netG(batch['noise'].clone().requires_grad_(True), tr.txt_embedding)
    errD = [float(tr.train_Dnet(i, 1)) for i in range(tr.num_Ds)]
    kl, errG_total = tr.train_Gnet(1)
    out[tag + '/errD'] = np.asarray(errD)
    out[tag + '/errG_total'], out[tag + '/kl'] = np.asarray(float(errG_total)), np.asarray(float(kl))
    out[tag + '/grad_emb'] = tr.txt_embedding.grad.numpy()
    for k, p_ in netG.named_parameters():
        if k in ('ca_net.fc.weight', 'h_net1.upsample4.1.weight', 'img_net1.img.0.weight', 'img_net2.img.0.weight',
                 'img_net3.img.0.weight', 'h_net3.upsample.1.weight'):
            out[tag + '/g_grad/' + k] = mg.sample(p_.grad)
    o = orc.train_step(ostate, batch, dims, uncond=float(cfg.TRAIN.COEFF.UNCOND_LOSS),
                       color_coeff=float(cfg.TRAIN.COEFF.COLOR_LOSS))
    assert np.allclose(o['errD'], errD, rtol=2e-4, atol=1e-5), (tag, o['errD'], errD)
    assert abs(o['errG_total'] - float(errG_total)) <= 2e-4 * abs(float(errG_total)) + 1e-5, (tag, o['errG_total'])
    assert torch.allclose(o['grad_emb'], tr.txt_embedding.grad, rtol=1e-3, atol=1e-6), tag
    for k, p_ in netG.named_parameters():
        g = o['grad_g'][k]
        assert float((g - p_.grad).norm()) <= 1e-4 * float(p_.grad.norm()) + 1e-9, (tag, k)


def main():
    torch.set_num_threads(8)
    cfg, rmodel, rtrainer = mg.import_reference()
    sys.path.insert(0, ROOT)
    from oracle import stackgan_oracle as orc
    out = {}
    c = CASE
    # ---- colour-consistency loss on ---------------------------------------------------------------------------
    mg.set_cfg(cfg, c)
    cfg.TRAIN.COEFF.COLOR_LOSS = 1.0
    _one_iteration(cfg, rmodel, rtrainer, c, 'color', out)
    # the reference's compute_mean_covariance on a fixed image batch
    g = torch.Generator().manual_seed(11)
    img = torch.rand(3, 3, 8, 16, generator=g) * 2 - 1
    mu, cov = rtrainer.compute_mean_covariance(img)
    omu, ocov = orc.compute_mean_covariance(img)
    assert torch.allclose(mu, omu) and torch.allclose(cov, ocov)
    out['meancov/mu'], out['meancov/cov'] = mu.numpy(), cov.numpy()
    # ---- unconditional loss off ---------------------------------------------------------------------------------
    mg.set_cfg(cfg, c)
    cfg.TRAIN.COEFF.UNCOND_LOSS = 0.0
    _one_iteration(cfg, rmodel, rtrainer, c, 'nouncond', out)
    # ---- B_CONDITION = False: model level -----------------------------------------------------------------------
    mg.set_cfg(cfg, c)
    cfg.GAN.B_CONDITION = False
    netG, netsD = mg.build_reference_nets(rmodel, rtrainer, c)
    out['nocond/g_keys'] = np.asarray(list(netG.state_dict().keys()))
    out['nocond/g_checksum'] = mg.checksum(netG.state_dict())
    for i, d in enumerate(netsD):
        out['nocond/d%d_keys' % i] = np.asarray(list(d.state_dict().keys()))
        out['nocond/d%d_checksum' % i] = mg.checksum(d.state_dict())
    batch = mg.make_batch(c)
    z = batch['noise'].clone().requires_grad_(True)
    fakes, mu, logvar = netG(z, None)
    assert mu is None and logvar is None
    dims = orc.Dims(c['branch'], c['gf'], c['df'], c['ef'], c['z'], c['t'], 2)
    ofakes, _, _ = orc.g_forward_nocond({k: v.clone() for k, v in netG.state_dict().items()}, batch['noise'], dims)
    total = 0
    gr = torch.Generator().manual_seed(5)
    for i, (f, d) in enumerate(zip(fakes, netsD)):
        assert torch.allclose(f, ofakes[i], rtol=1e-4, atol=1e-5), 'oracle != reference (nocond img %d)' % i
        out['nocond/fake%d_sample' % i] = mg.sample(f, 16384)
        logits, feat = d(f, None)
        assert len(logits) == 1
        ol, ofeat = orc.d_forward_nocond({k: v.clone() for k, v in d.state_dict().items()}, 64 << i, f.detach())
        assert torch.allclose(logits[0], ol[0], rtol=1e-4, atol=1e-5)
        out['nocond/d%d_logit' % i] = logits[0].detach().numpy()
        out['nocond/d%d_feat_sample' % i] = mg.sample(feat)
        r = torch.randn(feat.shape, generator=gr) * 0.01
        total = total + logits[0].sum() + (feat * r).sum()
    total.backward()
    out['nocond/grad_z'] = z.grad.numpy()
    for k, p_ in netG.named_parameters():
        if k in ('h_net1.fc.0.weight', 'h_net2.jointConv.0.weight', 'h_net3.jointConv.0.weight', 'img_net3.img.0.weight'):
            out['nocond/g_grad/' + k] = mg.sample(p_.grad)
    cfg.GAN.B_CONDITION = True
    np.savez_compressed(os.path.join(HERE, 'variants.npz'), **out)
    print('wrote variants', os.path.getsize(os.path.join(HERE, 'variants.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
