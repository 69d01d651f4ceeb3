"""Train-step harness with the surface of the reference's StackGAN_v2/trainer.py, on the MI355X kernels.

Same names and meaning as the reference where a caller can see them:
  weights_init (:65-75), load_params / copy_G_params (:78-85), KL_loss (:54-58),
  class_aware_loss (:298-311), load_network (:162-233), define_optimizers (:236-252),
  save_model (:255-265), condGANTrainer(output_dir, data_loader, imsize, my_dataset_flag,
  local_rank, distributed) with prepare_data / train_Dnet / train_Gnet / train (:318-638).

What is built differently (MI355X-first):
  * every network's parameters, gradients and Adam moments live in ONE flat fp32 buffer each, so the
    optimiser is a single fused kernel per network (Adam, plus the EMA of G), and data-parallel
    training is one RCCL all-reduce per network over xGMI instead of DDP's bucket stream;
  * the three D updates are independent, so D_i's all-reduce overlaps D_{i+1}'s forward/backward;
  * the G update does not compute (and never reduces) the discriminators' weight gradients the
    reference's backward produces and then discards (trainer.py:385 zeroes them);
  * per-replica BatchNorm statistics and per-replica class-aware loss, as in the reference's DDP.
The Inception-v3 forwards, IS/FID and image dumps of the reference loop are evaluation-side and out
of scope; the loop here runs the update and the checkpoint layout only.
"""
import os
import time
from copy import deepcopy

import torch
import torch.nn as nn

from . import ops
from .miscc.config import cfg
from .miscc.utils import mkdir_p
from .model import D_NET64, D_NET128, D_NET256, D_NET512, D_NET1024, G_NET, INCEPTION_V3  # noqa: F401


# ---- shared functions --------------------------------------------------------------------------------
def compute_mean_covariance(img):
    """Per-image channel mean (B,C,1,1) and C x C covariance over pixels (trainer.py:34-51).  Dormant in every
    BASELINE config (COLOR_LOSS = 0): plain torch ops on 3-channel images, no kernel of its own."""
    batch_size, channel_num, height, width = img.shape
    num_pixels = height * width
    mu = img.mean(2, keepdim=True).mean(3, keepdim=True)
    img_hat = (img - mu.expand_as(img)).view(batch_size, channel_num, num_pixels)
    covariance = torch.bmm(img_hat, img_hat.transpose(1, 2)) / num_pixels
    return mu, covariance


def KL_loss(mu, logvar):
    return ops.KLLoss.apply(mu, logvar)


def weights_init(m):
    """Conv*/Linear: orthogonal(gain 1); BatchNorm*: weight ~ N(1, 0.02), bias 0 (trainer.py:65-75)."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        nn.init.orthogonal_(m.weight.data, 1.0)
    elif classname.find('BatchNorm') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)
    elif classname.find('Linear') != -1:
        nn.init.orthogonal_(m.weight.data, 1.0)
        if m.bias is not None:
            m.bias.data.fill_(0.0)


def load_params(model, new_param):
    for p, new_p in zip(model.parameters(), new_param):
        p.data.copy_(new_p)
    ops.refresh_packed(model.parameters())


def copy_G_params(model):
    return deepcopy(list(p.data for p in model.parameters()))


def class_labels_to_device(class_labels, device):
    if torch.is_tensor(class_labels):
        return class_labels.to(device=device, dtype=torch.int32).contiguous()
    return torch.tensor([int(v) for v in class_labels], dtype=torch.int32, device=device)


def class_aware_loss(x_activates, class_labels):
    """x_activates (B, D); class_labels: sequence / tensor of B class ids (trainer.py:298-311)."""
    return ops.ClassAwareLoss.apply(x_activates, class_labels_to_device(class_labels, x_activates.device))


def _unwrap(net):
    return net.module if hasattr(net, 'module') else net


class _Replica(nn.Module):
    """Stand-in for the reference's DataParallel / DDP wrapper: same `module.`-prefixed state_dict
    keys (trainer.py:167-171, 192-196, 255-265) and a `device_ids` attribute, no comm of its own —
    gradients are reduced per network by FlatNet below."""

    def __init__(self, module, device_ids):
        super().__init__()
        self.module = module
        self.device_ids = list(device_ids)

    def forward(self, *a, **k):
        return self.module(*a, **k)


def load_network(gpus, distributed):
    """Build G and the per-scale Ds, initialise, wrap, optionally resume (trainer.py:162-233).
    The reference also builds INCEPTION_V3 here (a network download); the returned slot is None."""
    dev = torch.device('cuda', gpus[0]) if cfg.CUDA else torch.device('cpu')
    netG = G_NET()
    netG.apply(weights_init)
    netG = _Replica(netG.to(dev), gpus)
    classes = (D_NET64, D_NET128, D_NET256, D_NET512, D_NET1024)
    netsD = []
    for i in range(min(cfg.TREE.BRANCH_NUM, len(classes))):
        d = classes[i]()
        d.apply(weights_init)
        netsD.append(_Replica(d.to(dev), gpus))
    count = 0
    if cfg.TRAIN.NET_G != '':
        state_dict = torch.load(cfg.TRAIN.NET_G, map_location='cpu', weights_only=True)
        netG.load_state_dict(state_dict)
        istart = cfg.TRAIN.NET_G.rfind('_') + 1
        iend = cfg.TRAIN.NET_G.rfind('.')
        count = int(cfg.TRAIN.NET_G[istart:iend]) + 1
    if cfg.TRAIN.NET_D != '':
        for i in range(len(netsD)):
            state_dict = torch.load('%s%d.pth' % (cfg.TRAIN.NET_D, i), map_location='cpu', weights_only=True)
            netsD[i].load_state_dict(state_dict)
    return netG, netsD, len(netsD), None, count


# ---- flat parameter storage + fused optimiser -------------------------------------------------------------
class FlatNet:
    """All parameters of one network re-homed into a flat buffer (views keep nn.Module semantics);
    gradients accumulate into a second flat buffer; Adam moments in two more."""

    def __init__(self, net, lr, betas=(0.5, 0.999), eps=1e-8, with_ema=False):
        self.net = net
        self.params = [p for p in net.parameters()]
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.offsets, off = [], 0
        for n in self.sizes:
            self.offsets.append(off)
            off += (n + 3) & ~3  # keep every tensor 16-byte aligned inside the flat buffer
        self.total = off
        self.p = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.g = torch.zeros_like(self.p)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        for p, o, n in zip(self.params, self.offsets, self.sizes):
            self.p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.p[o:o + n].view_as(p)
            p.grad = self.g[o:o + n].view_as(p)
        self.avg = self.p.clone() if with_ema else None
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.step_count = 0
        ops.refresh_packed(self.params)

    def zero_grad(self):
        self.g.zero_()
        for p, o, n in zip(self.params, self.offsets, self.sizes):
            if p.grad is None or p.grad.data_ptr() != self.g.data_ptr() + 4 * o:
                p.grad = self.g[o:o + n].view_as(p)

    def set_requires_grad(self, flag):
        for p in self.params:
            p.requires_grad_(flag)

    def adam(self, gscale=1.0):
        ops.increment(self.step_dev)
        self.step_count += 1
        ops.adam_step(self.p, self.g, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps,
                      step_dev=self.step_dev, gscale=gscale)
        ops.refresh_packed(self.params)

    def ema(self, decay=0.999):
        ops.ema_update(self.avg, self.p, decay)

    def avg_params(self):
        return [self.avg[o:o + n].view_as(p) for p, o, n in zip(self.params, self.offsets, self.sizes)]


class _LegacyOptimizer:
    """`optimizer.step()` / `.zero_grad()` facade over FlatNet for callers written against
    torch.optim (define_optimizers' return values)."""

    def __init__(self, flat):
        self.flat = flat

    def step(self):
        # torch's own `net.zero_grad()` (set_to_none) detaches the parameters' .grad views from the flat gradient
        # buffer; stepping then would apply stale zeros.  Refuse instead of silently not training.
        f = self.flat
        for p, o in zip(f.params, f.offsets):
            if p.grad is None or p.grad.data_ptr() != f.g.data_ptr() + 4 * o:
                raise RuntimeError("optimizer.step(): a parameter's .grad no longer aliases the flat gradient buffer "
                                   "(was net.zero_grad(set_to_none=True) called?); use optimizer.zero_grad()")
        self.flat.adam()

    def zero_grad(self):
        self.flat.zero_grad()


def define_optimizers(netG, netsD):
    """Adam(lr from cfg, betas (0.5, 0.999)) per network (trainer.py:236-252), fused and flat."""
    flatsD = [FlatNet(_unwrap(d), cfg.TRAIN.DISCRIMINATOR_LR) for d in netsD]
    flatG = FlatNet(_unwrap(netG), cfg.TRAIN.GENERATOR_LR, with_ema=True)
    return _LegacyOptimizer(flatG), [_LegacyOptimizer(f) for f in flatsD]


def save_model(netG, avg_param_G, netsD, epoch, model_dir):
    """Model/netG_<count>.pth holds the EMA weights, Model/netD<i>.pth the live ones; keys carry the
    wrapper's `module.` prefix (trainer.py:255-265).  Tensors are cloned so the flat buffers behind
    the parameter views are not serialised whole."""
    load_params(netG, avg_param_G)
    torch.save({k: v.clone() for k, v in netG.state_dict().items()}, '%s/netG_%d.pth' % (model_dir, epoch))
    for i, netD in enumerate(netsD):
        torch.save({k: v.clone() for k, v in netD.state_dict().items()}, '%s/netD%d.pth' % (model_dir, i))
    print('Save G/Ds models.')


# ---- the trainer -----------------------------------------------------------------------------------------------
_D_STREAMS = {}   # (device, number of discriminators, priorities) -> per-discriminator HIP streams, shared process-wide


class condGANTrainer(object):
    d_overlap_min = 1 << 20     # smallest tail (elements) of a discriminator's flat gradient worth a second all-reduce

    def __init__(self, output_dir, data_loader, imsize, my_dataset_flag, local_rank=0, distributed=False):
        self.my_dataset_flag = my_dataset_flag
        if output_dir is not None:
            if not cfg.TRAIN.FLAG:
                output_dir += "_eval"
            self.model_dir = os.path.join(output_dir, 'Model')
            self.image_dir = os.path.join(output_dir, 'Image')
            self.log_dir = os.path.join(output_dir, 'Log')
            for d in (self.model_dir, self.image_dir, self.log_dir):
                mkdir_p(d)
        self.gpus = [local_rank]
        self.batch_size = cfg.TRAIN.BATCH_SIZE
        self.max_epoch = cfg.TRAIN.MAX_EPOCH
        self.snapshot_interval = cfg.TRAIN.SNAPSHOT_INTERVAL
        self.data_loader = data_loader
        self.num_batches = len(data_loader) if data_loader is not None else 0
        self.distributed = distributed
        self.world = torch.distributed.get_world_size() if distributed else 1
        self._pending = []
        self.stack_d_passes = True
        # independent discriminator passes on separate HIP streams (S2I_D_STREAMS=0 turns it off)
        self.d_streams = os.environ.get("S2I_D_STREAMS", "1") == "1" and torch.cuda.is_available()
        self._side_streams = None
        # hipGraph replay of the single-GPU step (S2I_GRAPH=1 or enable_graph()): the step has no host synchronisation and
        # a device-side Adam counter, so after a few eager warm-up steps it is captured once and replayed
        self._graph = None
        self._labels_dev = None
        self._g_pass = []
        if os.environ.get("S2I_GRAPH", "0") == "1":
            self.enable_graph()

    def _make_d_streams(self):
        """One stream per discriminator, all at the default priority.  (Round 2 gave the largest discriminator's stream the
        high HIP priority for 0.3 ms per step.  Round 3 found what that costs: while a high-priority queue holds PENDING
        packets -- the next step's discriminator work waiting on an event -- every other queue's kernels run slower, so a
        host that runs more than one step ahead made the step 20 % slower: 38.7 vs 31.6 ms, all of it in the generator's
        single-stream pieces (tools/replay_pieces.py, profiles/r03_replay_regimes.txt).  That, not branch serialisation,
        was also why hipGraph replay looked slow.)  The streams are shared by every trainer of the process."""
        key = (torch.cuda.current_device(), self.num_Ds)
        if key not in _D_STREAMS:
            _D_STREAMS[key] = [torch.cuda.Stream() for i in range(self.num_Ds)]
        return _D_STREAMS[key]

    # -- set-up -------------------------------------------------------------------------------------------
    def build(self, netG=None, netsD=None, start_count=0):
        """Networks + flat optimisers.  `train()` calls this; benches and tests may pass their own nets."""
        if netG is None:
            netG, netsD, _, _, start_count = load_network(self.gpus, self.distributed)
        self.netG, self.netsD, self.num_Ds = netG, netsD, len(netsD)
        self.optimizerG, self.optimizersD = define_optimizers(self.netG, self.netsD)
        self.flatG = self.optimizerG.flat
        self.flatsD = [o.flat for o in self.optimizersD]
        if self.distributed:
            for f in [self.flatG] + self.flatsD:  # identical start on every rank (DDP's initial broadcast)
                torch.distributed.broadcast(f.p, 0)
                if f.avg is not None:
                    f.avg.copy_(f.p)
                ops.refresh_packed(f.params)
                # DDP(broadcast_buffers=True) (trainer.py:167, 192): a resumed rank-0 checkpoint propagates its
                # BatchNorm running statistics too
                for buf in f.net.buffers():
                    torch.distributed.broadcast(buf, 0)
            self.rank = torch.distributed.get_rank()
            self._install_g_overlap()
            self._install_d_overlap()
        self.avg_param_G = self.flatG.avg_params()
        return start_count

    def prepare_data(self, data):
        if self.my_dataset_flag:
            imgs, w_imgs, t_embedding = data["real_image"], data["wrong_image"], data["real_embedding"]
            class_labels = data.get("class_label", data.get("text"))
        else:
            imgs, w_imgs, t_embedding, _, class_labels = data
        dev = torch.device('cuda', self.gpus[0])
        vembedding = t_embedding.float().to(dev, non_blocking=True).requires_grad_()
        def to_dev(t):
            t = t.to(dev, non_blocking=True)
            # datasets built with device_normalize=True hand over uint8 HWC batches: ToTensor + Normalize on the GPU
            return ops.images_from_uint8_hwc(t.contiguous()) if t.dtype == torch.uint8 else t
        real_vimgs = [to_dev(imgs[i]) for i in range(self.num_Ds)]
        wrong_vimgs = [to_dev(w_imgs[i]) for i in range(self.num_Ds)]
        return imgs, real_vimgs, wrong_vimgs, vembedding, class_labels

    # -- communication ------------------------------------------------------------------------------------------
    def _reduce_async(self, flat, lo=0, hi=None):
        if not self.distributed:
            return None
        g = flat.g if (lo == 0 and hi is None) else flat.g[lo:hi]
        return torch.distributed.all_reduce(g, op=torch.distributed.ReduceOp.SUM, async_op=True)

    def _install_g_overlap(self):
        """G's gradient all-reduce in two contiguous chunks: everything behind INIT_STAGE_G.upsample1 in parameter
        order (upsample2..4, the image heads, stages 2 and 3: their backward is complete when the gradient of
        upsample1's output exists) is reduced while upsample1 / fc / ca_net -- 80 % of G's parameters, last in the
        backward -- are still computing; the head chunk follows the backward.  Same sums as one all-reduce."""
        g = _unwrap(self.netG)
        h1 = getattr(g, 'h_net1', None)
        self._g_split, self._g_tail_work = None, None
        if h1 is None or os.environ.get("S2I_G_OVERLAP", "1") != "1":
            return
        ids = {id(p): k for k, p in enumerate(self.flatG.params)}
        first_tail = ids.get(id(h1.upsample2[1].weight))
        if first_tail is None:
            return
        self._g_split = self.flatG.offsets[first_tail]

        def hook(grad):
            if self._g_hook_armed:
                self._g_hook_armed = False
                self._g_tail_work = self._reduce_async(self.flatG, self._g_split, None)
            return None
        h1.after_up1_hook = hook
        self._g_hook_armed = False

    def _install_d_overlap(self):
        """The same for the discriminators: the parameters behind img_code_s16 in parameter order (tower, jointConv, logit
        heads: 68 of D_NET256's 71 M parameters) have their gradients first in the backward; their all-reduce starts from a
        hook on img_code_s16's output and runs under the backward of the four image-side convolutions, which is most of a
        discriminator's backward time.  Same sums as one all-reduce (`S2I_D_OVERLAP=0`: one all-reduce after the backward)."""
        n = len(self.netsD)
        self._d_split, self._d_tail_work = [None] * n, [None] * n
        self._d_hook_armed, self._d_fires, self._d_expected = [False] * n, [0] * n, [1] * n
        if os.environ.get("S2I_D_OVERLAP", "1") != "1":
            return
        for idx, netD in enumerate(self.netsD):
            d = _unwrap(netD)
            flat = self.flatsD[idx]
            head = {id(p) for p in d.img_code_s16.parameters()}
            k = 0
            while k < len(flat.params) and id(flat.params[k]) in head:
                k += 1
            if k == 0 or k >= len(flat.params):
                continue
            split = flat.offsets[k]
            if flat.g.numel() - split < self.d_overlap_min:
                continue                                   # a tail below 4 MB is not worth a second collective
            self._d_split[idx] = split

            def hook(grad, idx=idx):
                if self._d_hook_armed[idx]:
                    self._d_fires[idx] += 1
                    if self._d_fires[idx] == self._d_expected[idx]:   # un-stacked passes: the last of the three
                        self._d_hook_armed[idx] = False
                        self._d_tail_work[idx] = self._reduce_async(self.flatsD[idx], self._d_split[idx], None)
                return None
            d.after_s16_hook = hook

    # -- D update (trainer.py:375-427) ----------------------------------------------------------------------------
    def _d_logits(self, idx):
        """Conditional/unconditional probabilities of the real, wrong and fake batches.  When the batch is a
        multiple of 8 (so that every layer's rows split into whole 128-row tiles per batch) the three passes run
        as ONE stacked forward with per-batch BatchNorm statistics; otherwise as the reference's three calls."""
        netD = self.netsD[idx]
        mu = self.mu.detach()
        B = mu.shape[0]
        stacked = self.stack_d_passes and B % 8 == 0
        if getattr(self, '_d_expected', None) is not None:
            self._d_expected[idx] = 1 if stacked else 3     # backward firings of the img_code_s16 hook
        if stacked:
            x = torch.cat((self.real_imgs[idx], self.wrong_imgs[idx], self.fake_imgs[idx].detach()), 0)
            logits, _ = _unwrap(netD)(x, mu.repeat(3, 1), groups=3, need_features=False)
            self._stacked_logits = logits
            return [[l[g * B:(g + 1) * B] for l in logits] for g in range(3)]
        self._stacked_logits = None
        return [netD(self.real_imgs[idx], mu)[0], netD(self.wrong_imgs[idx], mu)[0],
                netD(self.fake_imgs[idx].detach(), mu)[0]]

    def _d_loss(self, idx):
        u = cfg.TRAIN.COEFF.UNCOND_LOSS
        real_logits, wrong_logits, fake_logits = self._d_logits(idx)
        if self._stacked_logits is not None and len(self._stacked_logits) == 2 and u > 0:
            # the six BCE terms of trainer.py:394-409 on the stacked (real | wrong | fake) heads, one launch:
            # cond targets 1/0/0, uncond targets 1/1/0 (wrong pairs count as real for the uncond head, :401)
            dev = self._stacked_logits[0].device
            key = (str(dev), float(u))
            if getattr(self, '_bce_consts', (None,))[0] != key:
                self._bce_consts = (key, torch.tensor([1., 1., 0., 1., 0., 0.], device=dev),
                                    torch.tensor([1., u, 1., u, 1., u], device=dev))
            return ops.BCEMulti.apply(self._bce_consts[1], self._bce_consts[2], 3, *self._stacked_logits)
        errD_real = ops.BCELoss.apply(real_logits[0], 1.0, 1.0)
        errD_wrong = ops.BCELoss.apply(wrong_logits[0], 0.0, 1.0)
        errD_fake = ops.BCELoss.apply(fake_logits[0], 0.0, 1.0)
        if len(real_logits) > 1 and u > 0:
            errD_real = errD_real + ops.BCELoss.apply(real_logits[1], 1.0, u)
            errD_wrong = errD_wrong + ops.BCELoss.apply(wrong_logits[1], 1.0, u)  # uncond target real (:401)
            errD_fake = errD_fake + ops.BCELoss.apply(fake_logits[1], 0.0, u)
            return errD_real + errD_wrong + errD_fake
        return errD_real + 0.5 * (errD_wrong + errD_fake)

    def train_Dnet(self, idx, count, defer_step=False):
        flat = self.flatsD[idx]
        flat.zero_grad()
        split = getattr(self, '_d_split', [None] * self.num_Ds)[idx] if self.distributed else None
        if split is not None:
            self._d_tail_work[idx], self._d_fires[idx] = None, 0
            self._d_hook_armed[idx] = True
        errD = self._d_loss(idx)
        errD.backward()
        if split is not None:
            self._d_hook_armed[idx] = False
        if split is not None and self._d_tail_work[idx] is not None:
            works = [self._reduce_async(flat, 0, split), self._d_tail_work[idx]]   # head chunk: img_code_s16
            self._d_tail_work[idx] = None
        else:
            works = [self._reduce_async(flat)]
        if defer_step:
            self._pending.append((flat, works))
        else:
            for work in works:
                if work is not None:
                    work.wait()
            flat.adam(1.0 / self.world)
        return errD

    def _flush_d_steps(self):
        for flat, works in self._pending:
            for work in works:
                if work is not None:
                    work.wait()
            flat.adam(1.0 / self.world)
        self._pending = []

    # -- G update (trainer.py:429-489) ------------------------------------------------------------------------------
    def _g_pass_of_d(self, i):
        """Discriminator i's part of the G update, forward AND backward, on the current stream: the adversarial (and
        class-aware) loss of the fake image of scale i, and its gradients w.r.t. that image and mu.  The images and mu enter
        as detached leaves, so this piece of the backward belongs to this discriminator's stream alone; the generator's
        backward (`_g_backward`) continues from the two gradients.  Same chain rule as one backward over the summed loss
        (trainer.py:437-446, 487)."""
        u = cfg.TRAIN.COEFF.UNCOND_LOSS
        flat = self.flatsD[i]
        flat.set_requires_grad(False)   # D's weight gradients would be discarded (trainer.py:385)
        try:
            fake = self.fake_imgs[i].detach().requires_grad_(True)
            mu = self.mu.detach().requires_grad_(True)
            outputs, x_active = self.netsD[i](fake, mu)
            errG = ops.BCELoss.apply(outputs[0], 1.0, 1.0)
            if len(outputs) > 1 and u > 0:
                errG = errG + ops.BCELoss.apply(outputs[1], 1.0, u)
            if cfg.TRAIN.COEFF.CAL_LOSS > 0:
                if self._labels_dev is None:
                    self._labels_dev = class_labels_to_device(self.class_labels, x_active.device)
                errG = errG + ops.ClassAwareLoss.apply(x_active, self._labels_dev).reshape(())
            g_fake, g_mu = torch.autograd.grad(errG, [fake, mu], allow_unused=True)
        finally:
            flat.set_requires_grad(True)
        self._g_pass[i] = (errG.detach(), g_fake, g_mu)

    def _g_backward(self):
        """The generator's own part of the G update on the current stream (after every `_g_pass_of_d`): KL term, the
        backward from the image gradients, all-reduce, Adam."""
        self.flatG.zero_grad()
        errG_total = 0
        roots, grads = [], []
        g_mu_total = None
        for i in range(self.num_Ds):
            errG, g_fake, g_mu = self._g_pass[i]
            errG_total = errG_total + errG
            roots.append(self.fake_imgs[i])
            grads.append(g_fake)
            if g_mu is not None:
                g_mu_total = g_mu if g_mu_total is None else g_mu_total + g_mu
        self._g_pass = [None] * self.num_Ds
        extra = 0
        if cfg.TRAIN.COEFF.COLOR_LOSS > 0:
            # colour-consistency between neighbouring scales (trainer.py:455-478); dormant by default
            coef = cfg.TRAIN.COEFF.COLOR_LOSS
            for hi, lo in ((-1, -2), (-2, -3)):
                if self.num_Ds >= -lo:
                    mu1, cov1 = compute_mean_covariance(self.fake_imgs[hi])
                    mu2, cov2 = compute_mean_covariance(self.fake_imgs[lo].detach())
                    extra = extra + coef * nn.functional.mse_loss(mu1, mu2) + coef * 5 * nn.functional.mse_loss(cov1, cov2)
        kl_loss = KL_loss(self.mu, self.logvar) * cfg.TRAIN.COEFF.KL
        own = kl_loss + extra
        errG_total = errG_total + own.detach()
        if g_mu_total is not None:
            roots.append(self.mu)
            grads.append(g_mu_total)
        roots.append(own)
        grads.append(torch.ones_like(own))
        split = getattr(self, '_g_split', None) if self.distributed else None
        self._g_tail_work = None
        self._g_hook_armed = split is not None
        torch.autograd.backward(roots, grads)
        self._g_hook_armed = False
        if self.distributed and self._g_tail_work is not None:
            work = self._reduce_async(self.flatG, 0, split)   # head chunk: ca_net, fc, upsample1
            work.wait()
            self._g_tail_work.wait()
            self._g_tail_work = None
        else:
            work = self._reduce_async(self.flatG)
            if work is not None:
                work.wait()
        self.flatG.adam(1.0 / self.world)
        return kl_loss.detach(), errG_total

    def train_Gnet(self, count):
        """One generator update: every discriminator's pass on its own HIP stream (forward, losses, backward to the fake
        image), then the generator's backward and Adam on the current stream."""
        self._labels_dev = None
        self._g_pass = [None] * self.num_Ds
        main = torch.cuda.current_stream() if self.d_streams else None
        if self.d_streams and self._side_streams is None:
            self._side_streams = self._make_d_streams()
        for i in range(self.num_Ds):
            if self.d_streams:
                self._side_streams[i].wait_stream(main)  # fake images / mu; D_i's own update is already in order
                with torch.cuda.stream(self._side_streams[i]):
                    self._g_pass_of_d(i)
            else:
                self._g_pass_of_d(i)
        if self.d_streams:
            for i in range(self.num_Ds):
                main.wait_stream(self._side_streams[i])
        return self._g_backward()

    # -- one iteration (trainer.py:536-572), Inception forwards excluded --------------------------------------------
    def enable_graph(self, warmup=3, executor="plan"):
        """Replay the single-GPU step from a recording after `warmup` eager steps.  The step is cut where its HIP streams
        fork and join, and every piece is captured as a graph of ONE stream: the generator forward (main stream), per
        discriminator its update followed by its pass of the G update (that discriminator's stream), the generator's
        backward + Adam + EMA (main stream).  The pieces are launched like kernels -- stream waits between them -- so the
        three discriminator streams stay concurrent.  executor="plan" (default): the captured graphs are only the RECORDING;
        their kernel nodes are re-issued as plain launches from one C call per piece (s2i_plan_replay, include/s2i_hip.h):
        hipGraphLaunch itself is slower than the Python step it replaces on ROCm 7.2.
        executor="graph": hipGraphLaunch of each piece.  Single process only: an RCCL all-reduce inside a recording is
        not exercised here."""
        # warm-up and capture run on one private stream: autograd's AccumulateGrad nodes remember the stream they were
        # created on, and one that lives on the (non-capturing) default stream invalidates the capture
        if executor not in ("plan", "graph"):
            raise ValueError("executor must be 'plan' or 'graph'")
        self._graph = dict(warmup=warmup, seen=0, graphs=None, stream=torch.cuda.Stream(), executor=executor)

    def _graph_signature(self, real_imgs, wrong_imgs, txt_embedding, noise, eps):
        ts = list(real_imgs) + list(wrong_imgs) + [txt_embedding, noise] + ([eps] if eps is not None else [])
        return tuple((tuple(t.shape), t.dtype) for t in ts) + (ops.ACT_BF16, ops.MATH_PLANES, bool(txt_embedding.requires_grad))

    def _capture(self, real_imgs, wrong_imgs, txt_embedding, class_labels, noise, eps):
        st = self._graph
        dev = noise.device
        static = dict(real=[t.detach().clone() for t in real_imgs], wrong=[t.detach().clone() for t in wrong_imgs],
                      emb=txt_embedding.detach().clone().requires_grad_(txt_embedding.requires_grad),
                      noise=noise.detach().clone(), eps=None if eps is None else eps.detach().clone(),
                      labels=class_labels_to_device(class_labels, dev).clone())
        # no autograd graph of an earlier iteration may survive into the capture (it would keep its AccumulateGrad nodes)
        self.fake_imgs = self.mu = self.logvar = self._stacked_logits = None
        self.real_imgs = self.wrong_imgs = self.txt_embedding = None
        import gc
        gc.collect()
        torch.cuda.synchronize()
        ops.pin_graph_resources()      # workspaces and pack tables referenced by the graphs are never freed from here on
        main = st['stream']
        if self._side_streams is None:
            self._side_streams = self._make_d_streams()
        n = self.num_Ds
        # one memory pool per stream: pieces that replay concurrently must not share freed blocks
        pools = [torch.cuda.graph_pool_handle() for _ in range(n + 1)]
        keep = st['executor'] == "plan"       # the plan reads the captured graph's nodes: keep the hipGraph_t
        graphs = dict(fwd=torch.cuda.CUDAGraph(keep_graph=keep), d=[torch.cuda.CUDAGraph(keep_graph=keep) for _ in range(n)],
                      g=torch.cuda.CUDAGraph(keep_graph=keep))
        errs = [None] * n
        with ops.param_grad_mode(True):
            with torch.cuda.graph(graphs['fwd'], pool=pools[n], stream=main):
                self._begin_step(static['real'], static['wrong'], static['emb'], static['labels'])
                self.fake_imgs, self.mu, self.logvar = _unwrap(self.netG)(static['noise'], static['emb'], static['eps'])
            for i in reversed(range(n)):
                with torch.cuda.graph(graphs['d'][i], pool=pools[i], stream=self._side_streams[i]):
                    errs[i] = self.train_Dnet(i, 0).detach()
                    self._g_pass_of_d(i)
            with torch.cuda.graph(graphs['g'], pool=pools[n], stream=main):
                kl_loss, errG_total = self._g_backward()
                self.flatG.ema(0.999)
                errD_total = 0
                for e in reversed(errs):        # the eager step's order: largest discriminator first
                    errD_total = errD_total + e
                outs = [o.detach().reshape(()) for o in (errD_total, errG_total, kl_loss)]
        for f in [self.flatG] + self.flatsD:
            f.step_count -= 1          # FlatNet.adam counted a step the capture did not execute; the replay below counts it
        plans = None
        if keep:
            import ctypes
            from . import _lib
            lib = _lib.load()

            def make_plan(g):
                handle, counts = ctypes.c_void_p(), (ctypes.c_int * 3)()
                _lib.check(lib.s2i_plan_create(g.raw_cuda_graph(), ctypes.byref(handle), counts), "s2i_plan_create")
                return handle, tuple(counts)
            plans = dict(fwd=make_plan(graphs['fwd']), d=[make_plan(g) for g in graphs['d']], g=make_plan(graphs['g']))
            st['launches'] = (plans['fwd'][1][0] + sum(p[1][0] for p in plans['d']) + plans['g'][1][0])
        st.update(graphs=graphs, plans=plans, static=static, outs=outs, pools=pools,
                  sig=self._graph_signature(real_imgs, wrong_imgs, txt_embedding, noise, eps))
        # the capture itself did not execute anything: replay once so that this call IS a step
        return self._replay(real_imgs, wrong_imgs, txt_embedding, class_labels, noise, eps, fresh=True)

    def _replay(self, real_imgs, wrong_imgs, txt_embedding, class_labels, noise, eps, fresh=False):
        st = self._graph
        s = st['static']

        def put(dst, src):
            if dst.data_ptr() != src.data_ptr():
                with torch.no_grad():
                    dst.copy_(src.detach(), non_blocking=True)
        if not fresh:
            for d, r in zip(s['real'], real_imgs):
                put(d, r)
            for d, r in zip(s['wrong'], wrong_imgs):
                put(d, r)
            put(s['emb'], txt_embedding)
            put(s['noise'], noise)
            if eps is not None:
                put(s['eps'], eps)
            put(s['labels'], class_labels_to_device(class_labels, noise.device))
        main = torch.cuda.current_stream()
        if st['plans'] is not None:
            from . import _lib
            lib, pl = _lib.load(), st['plans']
            _lib.check(lib.s2i_plan_replay(pl['fwd'][0], main.cuda_stream), "s2i_plan_replay")
            for i in reversed(range(self.num_Ds)):    # largest first, as the eager step enqueues them
                side = self._side_streams[i]
                side.wait_stream(main)
                _lib.check(lib.s2i_plan_replay(pl['d'][i][0], side.cuda_stream), "s2i_plan_replay")
            for i in range(self.num_Ds):
                main.wait_stream(self._side_streams[i])
            _lib.check(lib.s2i_plan_replay(pl['g'][0], main.cuda_stream), "s2i_plan_replay")
        else:
            g = st['graphs']
            g['fwd'].replay()
            for i in reversed(range(self.num_Ds)):
                side = self._side_streams[i]
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    g['d'][i].replay()
            for i in range(self.num_Ds):
                main.wait_stream(self._side_streams[i])
            g['g'].replay()
        for f in [self.flatG] + self.flatsD:
            f.step_count += 1
        if txt_embedding.requires_grad and s['emb'].grad is not None:
            txt_embedding.grad = s['emb'].grad.clone()     # the static gradient buffer is overwritten by the next replay
        return tuple(st['outs'])

    def train_step(self, real_imgs, wrong_imgs, txt_embedding, class_labels, noise, eps=None):
        st = self._graph
        if st is not None and not self.distributed and torch.cuda.is_available() and self.d_streams:
            if st['graphs'] is not None:
                if st['sig'] == self._graph_signature(real_imgs, wrong_imgs, txt_embedding, noise, eps):
                    return self._replay(real_imgs, wrong_imgs, txt_embedding, class_labels, noise, eps)
                # another batch shape (ragged last batch of an epoch): an eager step.  The replays bypassed the host-side
                # bookkeeping of the derived weight copies (bf16 / split planes, keyed by plan layout, and another batch
                # plans other layouts): drop them all, they are re-derived from the current masters on first use.
                for f in [self.flatG] + self.flatsD:
                    ops.invalidate_derived(f.params)
            elif st['seen'] >= st['warmup']:
                return self._capture(real_imgs, wrong_imgs, txt_embedding, class_labels, noise, eps)
            else:
                st['seen'] += 1
                gs, cur = st['stream'], torch.cuda.current_stream()
                gs.wait_stream(cur)
                with torch.cuda.stream(gs), ops.param_grad_mode(True):
                    out = self._train_step(real_imgs, wrong_imgs, txt_embedding, class_labels, noise, eps)
                cur.wait_stream(gs)
                return out
        # kernels accumulate into the flat gradient buffers (zeroed per update); the switch is scoped to the step, so
        # a later stock-optimiser use of the modules in this process gets autograd-returned gradients again.
        with ops.param_grad_mode(True):
            return self._train_step(real_imgs, wrong_imgs, txt_embedding, class_labels, noise, eps)

    def _begin_step(self, real_imgs, wrong_imgs, txt_embedding, class_labels):
        self.real_imgs, self.wrong_imgs = real_imgs, wrong_imgs
        self.txt_embedding, self.class_labels = txt_embedding, class_labels
        # class labels on the device once per step, on the main stream, before the discriminator streams fork
        self._labels_dev = (class_labels_to_device(class_labels, txt_embedding.device)
                            if cfg.TRAIN.COEFF.CAL_LOSS > 0 else None)
        self._g_pass = [None] * self.num_Ds

    def _train_step(self, real_imgs, wrong_imgs, txt_embedding, class_labels, noise, eps=None):
        self._begin_step(real_imgs, wrong_imgs, txt_embedding, class_labels)
        self.fake_imgs, self.mu, self.logvar = _unwrap(self.netG)(noise, txt_embedding, eps)
        errD_total = 0
        if self.d_streams and self.num_Ds >= 2:
            # the D updates are independent: each runs on its own HIP stream (zero, stacked forward, backward,
            # all-reduce, Adam, re-pack), so the many short kernels of one fill the idle CUs of the others.  The
            # same stream then carries that discriminator's pass of the G update (forward, losses, backward to the fake
            # image), which can start as soon as D_i is updated, while a larger discriminator is still in its own update.
            main = torch.cuda.current_stream()
            if self._side_streams is None:
                self._side_streams = self._make_d_streams()
            errs = []
            for i in reversed(range(self.num_Ds)):  # largest first: it is the critical path
                st = self._side_streams[i]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    errs.append(self.train_Dnet(i, 0))
            for i in range(self.num_Ds):
                with torch.cuda.stream(self._side_streams[i]):
                    self._g_pass_of_d(i)
            for i in range(self.num_Ds):
                main.wait_stream(self._side_streams[i])
        else:
            # largest first, so that its gradient all-reduce (285 MB for D_NET256) hides behind the smaller
            # discriminators' forward/backward
            errs = [self.train_Dnet(i, 0, defer_step=True) for i in reversed(range(self.num_Ds))]
            self._flush_d_steps()
            for i in range(self.num_Ds):
                self._g_pass_of_d(i)
        kl_loss, errG_total = self._g_backward()
        self.flatG.ema(0.999)
        for e in errs:
            errD_total = errD_total + e
        return errD_total, errG_total, kl_loss

    def train(self):
        start_count = self.build()
        dev = torch.device('cuda', self.gpus[0])
        nz = cfg.GAN.Z_DIM
        noise = torch.empty(self.batch_size, nz, device=dev)
        count = start_count
        start_epoch = start_count // max(self.num_batches, 1)
        errD_total = errG_total = kl_loss = None
        for epoch in range(start_epoch, self.max_epoch):
            start_t = time.time()
            for step, data in enumerate(self.data_loader, 0):
                _, real, wrong, emb, labels = self.prepare_data(data)
                noise.normal_(0, 1)
                errD_total, errG_total, kl_loss = self.train_step(real, wrong, emb, labels, noise[:emb.shape[0]])
                if step == 0 and self.gpus[0] == 0:
                    print('[%d/%d][%d/%d] Loss_D: %.2f Loss_G: %.2f'
                          % (epoch, self.max_epoch, step, self.num_batches, errD_total.item(), errG_total.item()))
                count += 1
                if count % cfg.TRAIN.SNAPSHOT_INTERVAL == 0:
                    self.save(count)
            if errD_total is not None and self.gpus[0] == 0:
                print('[%d/%d][%d] Loss_D: %.2f Loss_G: %.2f Loss_KL: %.2f Time: %.2fs'
                      % (epoch, self.max_epoch, self.num_batches, errD_total.item(), errG_total.item(),
                         kl_loss.item(), time.time() - start_t))
        self.save(count)

    def save(self, count):
        """The reference overwrites G's live weights with the EMA copy when it saves and never restores
        them (trainer.py:256, 590-601); the live weights are kept here.  With data-parallel ranks only
        rank 0 writes (the reference lets every rank write the same file names: a race on a shared filesystem); its
        BatchNorm buffers are the ones kept, as DDP's broadcast_buffers would leave them."""
        if self.distributed and torch.distributed.get_rank() != 0:
            return
        live = self.flatG.p.clone()
        save_model(self.netG, self.avg_param_G, self.netsD, count, self.model_dir)
        self.flatG.p.copy_(live)
        ops.refresh_packed(self.flatG.params)

    # -- evaluation (trainer.py:664-679, 681-803): images only; the Inception metrics are short-circuited upstream too
    def save_singleimages(self, images_u8, filenames, save_dir, split_dir, sentenceID, imsize, sample_idx=0):
        """images_u8: (B,H,W,3) uint8 on the host.  Same file naming as the reference."""
        from PIL import Image
        for i in range(images_u8.shape[0]):
            s_tmp = '%s/single_samples/%s/%s' % (save_dir, split_dir, filenames[i])
            folder = s_tmp[:s_tmp.rfind('/')]
            if not os.path.isdir(folder):
                mkdir_p(folder)
            Image.fromarray(images_u8[i]).save('%s_%d_sentence%d_%d.png' % (s_tmp, imsize, sentenceID, sample_idx))

    @torch.no_grad()
    def evaluate(self, split_dir):
        """G in eval mode over every embedding of every test item -> PNGs under <NET_G dir>/iteration<N>/.
        BatchNorm uses the running statistics (no batch barrier); the [-1,1] -> uint8 HWC conversion is one
        kernel on the NHWC output.  Returns the reference's placeholder metrics (trainer.py:803)."""
        if cfg.TRAIN.NET_G == '':
            print('Error: the path for morels is not found!')
            return None
        if split_dir == 'test':
            split_dir = 'valid'
        dev = torch.device('cuda', self.gpus[0])
        netG = G_NET()
        netG.apply(weights_init)
        netG = _Replica(netG.to(dev), self.gpus)
        netG.load_state_dict(torch.load(cfg.TRAIN.NET_G, map_location='cpu', weights_only=True))
        s_tmp = cfg.TRAIN.NET_G
        iteration = int(s_tmp[s_tmp.rfind('_') + 1:s_tmp.rfind('.')])
        save_dir = '%s/iteration%d' % (s_tmp[:s_tmp.rfind('/')], iteration)
        netG.eval()
        nz = cfg.GAN.Z_DIM
        imsize = cfg.TREE.BASE_SIZE * (2 ** (cfg.TREE.BRANCH_NUM - 1))
        for data in self.data_loader:
            imgs, t_embeddings, filenames = data
            t_embeddings = t_embeddings.float().to(dev)
            batch_size = t_embeddings.shape[0]
            noise = torch.empty(batch_size, nz, device=dev)
            for i in range(t_embeddings.size(1)):
                noise.normal_(0, 1)
                fake_imgs, _, _ = netG.module(noise, t_embeddings[:, i, :].contiguous(), None, True)
                u8 = ops.images_to_uint8_hwc(fake_imgs[-1]).cpu().numpy()
                self.save_singleimages(u8, filenames, save_dir, split_dir, i, imsize, 0)
        return [{'mu': 0, 'sigma': 0}, {'mu': 0, 'sigma': 0}]
