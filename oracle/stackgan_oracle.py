"""CPU oracle for the StackGAN-v2 G/D train step — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product path (speech_to_image_translation_without_text_amd/) never does and has no CPU fallback.

It restates, as pure functions over a {state_dict key: tensor} mapping and stock torch fp32 CPU
ops, the algorithm of the reference's hot path:
  generator        StackGAN_v2/model.py:112-354   (GLU, upBlock, ResBlock, CA_NET, INIT/NEXT_STAGE_G, G_NET)
  discriminators   StackGAN_v2/model.py:358-672   (encode_image_by_16times, downBlock, D_NET64/128/256/512/1024)
  losses           StackGAN_v2/trainer.py:54-58 (KL_loss), :298-311 (class_aware_loss)
  D / G updates    StackGAN_v2/trainer.py:375-489 (train_Dnet, train_Gnet), :236-252 (Adam), :571-572 (EMA)
  colour loss      StackGAN_v2/trainer.py:34-51, 455-478 (compute_mean_covariance, like_mu / like_cov terms)
Parity pin: tests/golden/*.npz hold outputs of the reference itself (imported on CPU in the build
container by tests/golden/make_golden.py, which also asserts this file against it); this file is
checked against those vectors by tests/test_oracle_golden.py.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


class Dims:
    """The cfg values the networks read at construct time (miscc/config.py:51-69)."""

    def __init__(self, branch_num=3, gf_dim=64, df_dim=64, ef_dim=128, z_dim=100, t_dim=1024, r_num=2):
        self.branch_num, self.gf_dim, self.df_dim = branch_num, gf_dim, df_dim
        self.ef_dim, self.z_dim, self.t_dim, self.r_num = ef_dim, z_dim, t_dim, r_num


def glu(x):  # model.py:116-122
    nc = x.size(1) // 2
    return x[:, :nc] * torch.sigmoid(x[:, nc:])


def _bn(p, prefix, x, training=True):
    """nn.BatchNorm1d/2d with torch defaults; running statistics in `p` are updated in place."""
    rm, rv = p.get(prefix + '.running_mean'), p.get(prefix + '.running_var')
    y = F.batch_norm(x, rm, rv, p[prefix + '.weight'], p[prefix + '.bias'], training, BN_MOMENTUM, BN_EPS)
    if training and (prefix + '.num_batches_tracked') in p:
        p[prefix + '.num_batches_tracked'] += 1
    return y


def up_block(p, prefix, x, training=True):  # model.py:133-140
    x = F.interpolate(x, scale_factor=2, mode='nearest')
    x = F.conv2d(x, p[prefix + '.1.weight'], padding=1)
    return glu(_bn(p, prefix + '.2', x, training))


def res_block(p, prefix, x, training=True):  # model.py:153-169
    h = F.conv2d(x, p[prefix + '.block.0.weight'], padding=1)
    h = glu(_bn(p, prefix + '.block.1', h, training))
    h = F.conv2d(h, p[prefix + '.block.3.weight'], padding=1)
    return _bn(p, prefix + '.block.4', h, training) + x


def ca_net(p, emb, eps, ef_dim):  # model.py:182-200
    h = glu(F.linear(emb, p['ca_net.fc.weight'], p['ca_net.fc.bias']))
    mu, logvar = h[:, :ef_dim], h[:, ef_dim:]
    c = eps * torch.exp(0.5 * logvar) + mu
    return c, mu, logvar


def init_stage(p, prefix, z, c, ngf, training=True):  # model.py:227-244
    h = F.linear(torch.cat((c, z), 1), p[prefix + '.fc.0.weight'])
    h = glu(_bn(p, prefix + '.fc.1', h, training)).view(-1, ngf, 4, 4)
    for i in range(1, 5):
        h = up_block(p, '%s.upsample%d' % (prefix, i), h, training)
    return h


def next_stage(p, prefix, h, c, r_num, training=True):  # model.py:272-284
    s = h.size(2)
    cc = c.view(c.size(0), -1, 1, 1).repeat(1, 1, s, s)
    h = F.conv2d(torch.cat((cc, h), 1), p[prefix + '.jointConv.0.weight'], padding=1)
    h = glu(_bn(p, prefix + '.jointConv.1', h, training))
    for r in range(r_num):
        h = res_block(p, '%s.residual.%d' % (prefix, r), h, training)
    return up_block(p, prefix + '.upsample', h, training)


def get_image(p, prefix, h):  # model.py:287-298
    return torch.tanh(F.conv2d(h, p[prefix + '.img.0.weight'], padding=1))


def g_forward_nocond(p, z, dims, training=True):
    """G_NET.forward with cfg.GAN.B_CONDITION = False (model.py:308, 332-336, 229-233, 254-255): no ca_net, the noise
    itself is the code that NEXT_STAGE_G broadcasts; INIT_STAGE_G's fc sees z only.  -> ([imgs], None, None)."""
    h = F.linear(z, p['h_net1.fc.0.weight'])
    h = glu(_bn(p, 'h_net1.fc.1', h, training)).view(-1, dims.gf_dim * 16, 4, 4)
    for i in range(1, 5):
        h = up_block(p, 'h_net1.upsample%d' % i, h, training)
    imgs = [get_image(p, 'img_net1', h)]
    for i in range(2, dims.branch_num + 1):
        h = next_stage(p, 'h_net%d' % i, h, z, dims.r_num, training)
        imgs.append(get_image(p, 'img_net%d' % i, h))
    return imgs, None, None


def d_forward_nocond(p, size, x, training=True):
    """D_NETxx.forward with cfg.GAN.B_CONDITION = False (model.py:418, 430-445): no jointConv, one head."""
    h = F.leaky_relu(F.conv2d(x, p['img_code_s16.0.weight'], stride=2, padding=1), 0.2)
    for ci in (2, 5, 8):
        h = F.conv2d(h, p['img_code_s16.%d.weight' % ci], stride=2, padding=1)
        h = F.leaky_relu(_bn(p, 'img_code_s16.%d' % (ci + 1), h, training), 0.2)
    for name, stride in _D_TOWER[size]:
        h = _leaky_block(p, name, h, stride, training)
    out = torch.sigmoid(F.conv2d(h, p['logits.0.weight'], p['logits.0.bias'], stride=4)).view(-1)
    return [out], h.reshape(h.shape[0], -1)


def g_forward(p, z, emb, eps, dims, training=True):
    """G_NET.forward (model.py:327-354) -> ([img64, img128, img256], mu, logvar)."""
    c, mu, logvar = ca_net(p, emb, eps, dims.ef_dim)
    imgs = []
    h = init_stage(p, 'h_net1', z, c, dims.gf_dim * 16, training)
    imgs.append(get_image(p, 'img_net1', h))
    for i in range(2, dims.branch_num + 1):
        h = next_stage(p, 'h_net%d' % i, h, c, dims.r_num, training)
        imgs.append(get_image(p, 'img_net%d' % i, h))
    return imgs, mu, logvar


class MaskTape:
    """LeakyReLU decisions of one discriminator forward, in call order.  `record` mode stores `z > 0` of every
    LeakyReLU; `replay` mode applies the stored decisions instead of the sign of z.  Test instrument: the fused GPU
    path and this CPU path see pre-activations that differ by ~1e-6, so a z that close to zero can take different
    sides of the kink; replaying the GPU's decisions here isolates that effect from arithmetic differences."""

    def __init__(self, masks=None):
        self.masks = [] if masks is None else list(masks)
        self.replay = masks is not None
        self.pos = 0

    def lrelu(self, z):
        if not self.replay:
            self.masks.append((z > 0).detach())
            return F.leaky_relu(z, 0.2)
        m = self.masks[self.pos]
        self.pos += 1
        return torch.where(m, z, 0.2 * z)


def _lrelu(z, tape):
    return F.leaky_relu(z, 0.2) if tape is None else tape.lrelu(z)


def _leaky_block(p, prefix, x, stride, training=True, tape=None):  # model.py:358-376
    k = p[prefix + '.0.weight']
    x = F.conv2d(x, k, stride=stride, padding=1)
    return _lrelu(_bn(p, prefix + '.1', x, training), tape)


_D_TOWER = {
    64: (),
    128: (('img_code_s32', 2), ('img_code_s32_1', 1)),
    256: (('img_code_s32', 2), ('img_code_s64', 2), ('img_code_s64_1', 1), ('img_code_s64_2', 1)),
    # D_NET512 / D_NET1024 (model.py:555-613, 616-672): three / four downBlocks, then Block3x3_leakRelu back to 8 ndf
    512: (('img_code_s32', 2), ('img_code_s64', 2), ('img_code_s128', 2), ('img_code_s128_1', 1), ('img_code_s128_2', 1),
          ('img_code_s128_3', 1)),
    1024: (('img_code_s32', 2), ('img_code_s64', 2), ('img_code_s128', 2), ('img_code_s256', 2), ('img_code_s256_1', 1),
           ('img_code_s256_2', 1), ('img_code_s256_3', 1), ('img_code_s256_4', 1)),
}


def d_forward(p, size, x, c, training=True, tape=None):
    """D_NET64/128/256/512/1024.forward (model.py:424-445, 473-496, 526-551, 584-613, 648-672) -> ([cond, uncond], x_immediate).
    `tape` (MaskTape) records or replays the LeakyReLU decisions (test instrument)."""
    h = _lrelu(F.conv2d(x, p['img_code_s16.0.weight'], stride=2, padding=1), tape)  # model.py:383-384
    for ci in (2, 5, 8):
        h = F.conv2d(h, p['img_code_s16.%d.weight' % ci], stride=2, padding=1)
        h = _lrelu(_bn(p, 'img_code_s16.%d' % (ci + 1), h, training), tape)
    for name, stride in _D_TOWER[size]:
        h = _leaky_block(p, name, h, stride, training, tape)
    x_immediate = h.reshape(h.shape[0], -1)
    cc = c.view(c.size(0), -1, 1, 1).repeat(1, 1, 4, 4)
    hc = _leaky_block(p, 'jointConv', torch.cat((cc, h), 1), 1, training, tape)
    cond = torch.sigmoid(F.conv2d(hc, p['logits.0.weight'], p['logits.0.bias'], stride=4)).view(-1)
    uncond = torch.sigmoid(F.conv2d(h, p['uncond_logits.0.weight'], p['uncond_logits.0.bias'], stride=4)).view(-1)
    return [cond, uncond], x_immediate


def kl_loss(mu, logvar):  # trainer.py:54-58
    return torch.mean(1 + logvar - mu.pow(2) - logvar.exp()) * -0.5


def class_aware_loss(x, labels):  # trainer.py:298-311
    B, D = x.shape
    scores = x @ x.t()
    lab = torch.as_tensor([int(v) for v in labels])
    pair = (lab[:, None] == lab[None, :]) & ~torch.eye(B, dtype=torch.bool)
    if int(pair.sum()) > 0:
        return torch.clamp(scores.mean() - scores[pair].mean(), min=0).div(D).view(1)
    return torch.zeros(1)


def compute_mean_covariance(img):  # trainer.py:34-51
    B, C, H, W = img.shape
    mu = img.mean(2, keepdim=True).mean(3, keepdim=True)
    hat = (img - mu.expand_as(img)).view(B, C, H * W)
    return mu, torch.bmm(hat, hat.transpose(1, 2)) / (H * W)


def colour_loss(fakes, coeff):
    """Colour-consistency terms of train_Gnet (trainer.py:455-478): neighbouring scales' per-image channel means and
    covariances, the lower scale detached; the covariance term carries the extra factor 5."""
    total = 0.0
    for hi, lo in ((-1, -2), (-2, -3)):
        if len(fakes) >= -lo:
            mu1, cov1 = compute_mean_covariance(fakes[hi])
            mu2, cov2 = compute_mean_covariance(fakes[lo].detach())
            total = total + coeff * F.mse_loss(mu1, mu2) + coeff * 5 * F.mse_loss(cov1, cov2)
    return total


def bce(prob, target):  # nn.BCELoss(), trainer.py:499
    return F.binary_cross_entropy(prob, torch.full_like(prob, float(target)))


def adam_update(p, g, state, lr, betas=(0.5, 0.999), eps=1e-8):
    """torch.optim.Adam defaults as used at trainer.py:240-251 (no weight decay, no amsgrad)."""
    b1, b2 = betas
    state['step'] = state.get('step', 0) + 1
    t = state['step']
    m = state.setdefault('m', torch.zeros_like(p))
    v = state.setdefault('v', torch.zeros_like(p))
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def _trainable(p):
    return [k for k in p if not (k.endswith('running_mean') or k.endswith('running_var')
                                 or k.endswith('num_batches_tracked'))]


def _with_grad(p):
    q = dict(p)
    for k in _trainable(p):
        q[k] = p[k].detach().clone().requires_grad_(True)
    return q


class TrainState:
    """Weights, Adam moments and the EMA copy of G for the oracle's train step."""

    def __init__(self, g_params, d_params_list):
        self.g = {k: v.clone() for k, v in g_params.items()}
        self.ds = [{k: v.clone() for k, v in d.items()} for d in d_params_list]
        self.opt_g = {}
        self.opt_ds = [{} for _ in d_params_list]
        self.avg_g = {k: self.g[k].clone() for k in _trainable(self.g)}


def train_step(state, batch, dims, lr_g=2e-4, lr_d=2e-4, uncond=1.0, use_cal=True, kl_coeff=2.0, color_coeff=0.0):
    """One iteration of condGANTrainer.train's loop body (trainer.py:536-572), Inception excluded.

    batch: dict(emb, noise, eps, real=[...], wrong=[...], labels).  Returns a dict of losses and the
    fake images; `state` is updated in place.
    """
    sizes = [64, 128, 256][:dims.branch_num]
    # (1) generate fake images; the graph is kept for the G update (trainer.py:542-545)
    gp = _with_grad(state.g)
    emb = batch['emb'].detach().clone().requires_grad_(True)
    fakes, mu, logvar = g_forward(gp, batch['noise'], emb, batch['eps'], dims)
    out = {'fake': [f.detach().clone() for f in fakes], 'mu': mu.detach().clone(), 'logvar': logvar.detach().clone()}
    # (2) update each D (trainer.py:375-427)
    errD_total = 0.0
    out['errD'] = []
    for i, size in enumerate(sizes):
        dp = _with_grad(state.ds[i])
        c = mu.detach()
        real_l, _ = d_forward(dp, size, batch['real'][i], c)
        wrong_l, _ = d_forward(dp, size, batch['wrong'][i], c)
        fake_l, _ = d_forward(dp, size, fakes[i].detach(), c)
        if uncond > 0:
            errD = (bce(real_l[0], 1) + uncond * bce(real_l[1], 1)
                    + bce(wrong_l[0], 0) + uncond * bce(wrong_l[1], 1)      # wrong pairs: uncond target is REAL (:401)
                    + bce(fake_l[0], 0) + uncond * bce(fake_l[1], 0))
        else:  # trainer.py:411-412
            errD = bce(real_l[0], 1) + 0.5 * (bce(wrong_l[0], 0) + bce(fake_l[0], 0))
        keys = _trainable(dp)
        grads = torch.autograd.grad(errD, [dp[k] for k in keys], allow_unused=True)
        for k, g in zip(keys, grads):
            if g is None:  # unused head (UNCOND_LOSS = 0): torch.optim.Adam skips parameters without a gradient
                continue
            st = state.opt_ds[i].setdefault(k, {})
            adam_update(state.ds[i][k], g, st, lr_d)
        for k in state.ds[i]:
            if k not in keys:
                state.ds[i][k] = dp[k]  # running statistics advanced by the three forwards
        out['errD'].append(float(errD))
        errD_total += float(errD)
    # (3) update G through the updated Ds (trainer.py:429-489)
    errG_total = 0.0
    cal_total = torch.zeros(1)
    out['errG'] = []
    for i, size in enumerate(sizes):
        dp = dict(state.ds[i])
        logits, feat = d_forward(dp, size, fakes[i], mu)
        errG = bce(logits[0], 1)
        if uncond > 0:
            errG = errG + uncond * bce(logits[1], 1)
        if use_cal:  # the CAL_LOSS coefficient is only a switch (trainer.py:444-446)
            cal_total = cal_total + class_aware_loss(feat, batch['labels'])
        errG_total = errG_total + errG
        out['errG'].append(float(errG))
        for k in state.ds[i]:
            state.ds[i][k] = dp[k]
    if color_coeff > 0:
        errG_total = errG_total + colour_loss(fakes, color_coeff)
    kl = kl_loss(mu, logvar) * kl_coeff
    errG_total = errG_total + kl + cal_total
    keys = _trainable(gp)
    grads = torch.autograd.grad(errG_total, [gp[k] for k in keys] + [emb], allow_unused=True)
    out['grad_emb'] = grads[-1].detach().clone()
    out['grad_g'] = {k: g.detach().clone() for k, g in zip(keys, grads[:-1])}
    for k, g in zip(keys, grads[:-1]):
        st = state.opt_g.setdefault(k, {})
        adam_update(state.g[k], g, st, lr_g)
    for k in state.g:
        if k not in keys:
            state.g[k] = gp[k]
    # EMA (trainer.py:571-572)
    for k in state.avg_g:
        state.avg_g[k].mul_(0.999).add_(state.g[k], alpha=0.001)
    out.update(errD_total=errD_total, errG_total=float(errG_total), kl=float(kl), cal=float(cal_total))
    return out
