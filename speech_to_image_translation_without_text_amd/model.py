"""StackGAN-v2 generator / discriminators on the MI355X kernels, behind the module surface of the
reference's StackGAN_v2/model.py (G_NET :301-354, D_NET64/128/256 :402-551, D_NET512/1024 :555-672).

Drop-in contract:
  * zero-argument constructors reading the global `cfg` at construct time;
  * `G_NET()(z, emb) -> ([img64, img128, img256] NCHW in [-1,1], mu, logvar)`;
  * `D_NETxx()(img NCHW, c_code) -> ([cond, uncond] probabilities (B,), x_immediate (B, 8192))`;
  * identical `state_dict()` keys, shapes and order, identical `parameters()` order, sub-modules that
    `trainer.weights_init` recognises by class name (Conv2d / BatchNorm* / Linear), and identical
    consumption of torch's RNG at construction, so that seeded weights equal the reference's.

What differs is everything underneath: the containers below hold the reference's parameters but
never run torch.nn forward code.  Each block's forward is one fused operator from ops.py (HIP
kernels through the C-ABI), activations travel NHWC between blocks, the nearest-x2 upsample and the
c_code concat are folded into the convolution's gather, and images cross the module boundary as
NCHW.  There is no CPU path: calling forward on CPU tensors raises.
"""
import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_GLU, ACT_LRELU, ACT_NONE, ACT_TANH
from .miscc.config import cfg


# ---- parameter containers with fused forwards -------------------------------------------------------
class GLU(nn.Module):
    """Gated linear unit over the channel axis of a 2-D tensor (model.py:112-122)."""

    def forward(self, x):
        if x.size(1) % 2 != 0:
            raise AssertionError('channels dont divide 2!')
        return ops.Glu2d.apply(x)


def conv3x3(in_planes, out_planes):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=1, padding=1, bias=False)


def _bn_state(bn):
    return (bn.running_mean, bn.running_var, bn.num_batches_tracked)


class _Fused(nn.Sequential):
    """nn.Sequential used as a parameter container; `conv_at` / `bn_at` index the children that the
    reference's Sequential holds at the same positions (so state_dict keys match)."""
    kind = "k3s1"
    act = ACT_NONE
    conv_at = 0
    bn_at = 1

    def forward(self, x, cvec=None, residual=None, groups=1, defer=False):
        """defer=True: the caller hands the result to exactly ONE further convolution block, which may gather this block's
        raw conv output and apply BatchNorm + LeakyReLU while it loads (ops.py, "apply-on-load")."""
        conv, bn = self[self.conv_at], self[self.bn_at]
        return ops.ConvBnAct.apply(x, cvec, conv.weight, bn.weight, bn.bias, residual, self.kind, self.act,
                                   _bn_state(bn), self.training, groups, defer)


class UpBlock(_Fused):
    """nearest x2 -> conv3x3 -> BatchNorm -> GLU (model.py:133-140) as one 4-phase transposed conv."""
    kind, act, conv_at, bn_at = "up", ACT_GLU, 1, 2

    def __init__(self, in_planes, out_planes):
        super().__init__(nn.Upsample(scale_factor=2, mode='nearest'), conv3x3(in_planes, out_planes * 2),
                         nn.BatchNorm2d(out_planes * 2), GLU())


def upBlock(in_planes, out_planes):
    return UpBlock(in_planes, out_planes)


class Block3x3Relu(_Fused):
    """conv3x3 -> BatchNorm -> GLU (model.py:144-150)."""
    kind, act = "k3s1", ACT_GLU

    def __init__(self, in_planes, out_planes):
        super().__init__(conv3x3(in_planes, out_planes * 2), nn.BatchNorm2d(out_planes * 2), GLU())


def Block3x3_relu(in_planes, out_planes):
    return Block3x3Relu(in_planes, out_planes)


class ResBlock(nn.Module):
    """x + BN(conv(GLU(BN(conv(x))))) (model.py:153-169); the add rides the second BN-apply kernel."""

    def __init__(self, channel_num):
        super().__init__()
        self.block = nn.Sequential(conv3x3(channel_num, channel_num * 2), nn.BatchNorm2d(channel_num * 2), GLU(),
                                   conv3x3(channel_num, channel_num), nn.BatchNorm2d(channel_num))

    def forward(self, x):
        b = self.block
        h = ops.ConvBnAct.apply(x, None, b[0].weight, b[1].weight, b[1].bias, None, "k3s1", ACT_GLU, _bn_state(b[1]),
                                self.training)
        return ops.ConvBnAct.apply(h, None, b[3].weight, b[4].weight, b[4].bias, x, "k3s1", ACT_NONE, _bn_state(b[4]),
                                   self.training)


class Block3x3LeakRelu(_Fused):
    """conv3x3 -> BatchNorm -> LeakyReLU(0.2) (model.py:358-365)."""
    kind, act = "k3s1", ACT_LRELU

    def __init__(self, in_planes, out_planes):
        super().__init__(conv3x3(in_planes, out_planes), nn.BatchNorm2d(out_planes), nn.LeakyReLU(0.2, inplace=False))


def Block3x3_leakRelu(in_planes, out_planes):
    return Block3x3LeakRelu(in_planes, out_planes)


class DownBlock(_Fused):
    """Conv2d(k4,s2,p1) -> BatchNorm -> LeakyReLU(0.2) (model.py:369-376)."""
    kind, act = "k4s2", ACT_LRELU

    def __init__(self, in_planes, out_planes):
        super().__init__(nn.Conv2d(in_planes, out_planes, 4, 2, 1, bias=False), nn.BatchNorm2d(out_planes),
                         nn.LeakyReLU(0.2, inplace=False))


def downBlock(in_planes, out_planes):
    return DownBlock(in_planes, out_planes)


class EncodeImageBy16(nn.Sequential):
    """Four stride-2 stages, the first without BatchNorm (model.py:380-398).  Input: NHWC4 image."""

    def __init__(self, ndf):
        layers = [nn.Conv2d(3, ndf, 4, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=False)]
        for i in range(3):
            layers += [nn.Conv2d(ndf << i, ndf << (i + 1), 4, 2, 1, bias=False), nn.BatchNorm2d(ndf << (i + 1)),
                       nn.LeakyReLU(0.2, inplace=False)]
        super().__init__(*layers)

    def forward(self, x, groups=1, taps=None, defer_last=False):
        """defer_last: the result goes to exactly one further convolution block (the deeper discriminators' towers)."""
        h = ops.ConvAct.apply(x, self[0].weight, None, "k4s2", ACT_LRELU, self[0].out_channels)
        if taps is not None:
            taps.append(h)
        for ci in (2, 5, 8):
            bn = self[ci + 1]
            # inside the chain every block feeds only the next one: BatchNorm + LeakyReLU are applied by that block's gather
            defer = taps is None and (ci != 8 or defer_last)
            h = ops.ConvBnAct.apply(h, None, self[ci].weight, bn.weight, bn.bias, None, "k4s2", ACT_LRELU,
                                    _bn_state(bn), self.training, groups, defer)
            if taps is not None:
                taps.append(h)
        return h


def encode_image_by_16times(ndf):
    return EncodeImageBy16(ndf)


# ---- generator ------------------------------------------------------------------------------------------
class CA_NET(nn.Module):
    """Conditioning augmentation (model.py:172-200): Linear+GLU -> (mu, logvar) -> reparameterise."""

    def __init__(self):
        super().__init__()
        self.t_dim = cfg.TEXT.DIMENSION
        self.ef_dim = cfg.GAN.EMBEDDING_DIM
        self.fc = nn.Linear(self.t_dim, self.ef_dim * 4, bias=True)
        self.relu = GLU()

    def _encode(self, text_embedding):
        B = text_embedding.shape[0]
        pre = ops.ConvAct.apply(text_embedding.reshape(B, 1, 1, self.t_dim), self.fc.weight, self.fc.bias, "k1",
                                ACT_NONE, self.ef_dim * 4)
        h = ops.Glu2d.apply(pre.view(B, self.ef_dim * 4))
        return h[:, :self.ef_dim], h[:, self.ef_dim:], h

    def encode(self, text_embedding):
        mu, logvar, _ = self._encode(text_embedding)
        return mu, logvar

    def forward(self, text_embedding, eps=None):
        mu, logvar, h = self._encode(text_embedding)
        if eps is None:
            # same generator the reference draws from (model.py:190-193): torch's global RNG of the device
            eps = torch.empty_like(mu).normal_()
        return ops.Reparam.apply(h, eps), mu, logvar


class INIT_STAGE_G(nn.Module):
    """(c_code, z) -> fc/BN1d/GLU -> 4x4 map -> four up-blocks (model.py:203-244)."""

    def __init__(self, ngf):
        super().__init__()
        self.gf_dim = ngf
        self.in_dim = cfg.GAN.Z_DIM + (cfg.GAN.EMBEDDING_DIM if cfg.GAN.B_CONDITION else 0)
        self.fc = nn.Sequential(nn.Linear(self.in_dim, ngf * 4 * 4 * 2, bias=False), nn.BatchNorm1d(ngf * 4 * 4 * 2),
                                GLU())
        self.upsample1 = upBlock(ngf, ngf // 2)
        self.upsample2 = upBlock(ngf // 2, ngf // 4)
        self.upsample3 = upBlock(ngf // 4, ngf // 8)
        self.upsample4 = upBlock(ngf // 8, ngf // 16)

    def forward(self, z_code, c_code=None):
        B = z_code.shape[0]
        cvec = c_code if (cfg.GAN.B_CONDITION and c_code is not None) else None
        lin, bn = self.fc[0], self.fc[1]
        h = ops.ConvBnAct.apply(z_code.reshape(B, 1, 1, -1), cvec, lin.weight, bn.weight, bn.bias, None, "k1", ACT_GLU,
                                _bn_state(bn), self.training)
        h = ops.ToNHWC.apply(h.view(B, self.gf_dim, 4, 4), self.gf_dim)
        h = self.upsample1(h)
        hook = getattr(self, 'after_up1_hook', None)
        if hook is not None and h.requires_grad:
            # fires when the gradient of upsample1's output exists, i.e. when every later layer's parameter gradients
            # have been issued: the data-parallel trainer starts reducing that part of G's flat gradient there
            h.register_hook(hook)
        for up in (self.upsample2, self.upsample3, self.upsample4):
            h = up(h)
        return h


class NEXT_STAGE_G(nn.Module):
    """cat(c_code, h) -> jointConv -> residual blocks -> up-block (model.py:247-284)."""

    def __init__(self, ngf, num_residual=None):
        super().__init__()
        self.gf_dim = ngf
        self.ef_dim = cfg.GAN.EMBEDDING_DIM if cfg.GAN.B_CONDITION else cfg.GAN.Z_DIM
        self.num_residual = cfg.GAN.R_NUM if num_residual is None else num_residual
        self.jointConv = Block3x3_relu(ngf + self.ef_dim, ngf)
        self.residual = nn.Sequential(*[ResBlock(ngf) for _ in range(self.num_residual)])
        self.upsample = upBlock(ngf, ngf // 2)

    def forward(self, h_code, c_code):
        h = self.jointConv(h_code, cvec=c_code.reshape(-1, self.ef_dim))
        h = self.residual(h)
        return self.upsample(h)


class GET_IMAGE_G(nn.Module):
    """conv3x3(ngf -> 3) + tanh (model.py:287-298); NHWC in, NCHW image out."""

    def __init__(self, ngf):
        super().__init__()
        self.gf_dim = ngf
        self.img = nn.Sequential(conv3x3(ngf, 3), nn.Tanh())

    def forward(self, h_code, nhwc=False):
        img4 = ops.ConvAct.apply(h_code, self.img[0].weight, None, "k3s1", ACT_TANH, 4)
        return img4 if nhwc else ops.ToNCHW.apply(img4, 3)


class G_NET(nn.Module):
    def __init__(self):
        super().__init__()
        self.gf_dim = cfg.GAN.GF_DIM
        self.branch_num = cfg.TREE.BRANCH_NUM
        self.b_condition = cfg.GAN.B_CONDITION
        if self.b_condition:
            self.ca_net = CA_NET()
        n = self.branch_num
        if n > 0:
            self.h_net1 = INIT_STAGE_G(self.gf_dim * 16)
            self.img_net1 = GET_IMAGE_G(self.gf_dim)
        if n > 1:
            self.h_net2 = NEXT_STAGE_G(self.gf_dim)
            self.img_net2 = GET_IMAGE_G(self.gf_dim // 2)
        if n > 2:
            self.h_net3 = NEXT_STAGE_G(self.gf_dim // 2)
            self.img_net3 = GET_IMAGE_G(self.gf_dim // 4)
        if n > 3:  # untested upstream (model.py:320-325); kept so that the same keys exist
            self.h_net4 = NEXT_STAGE_G(self.gf_dim // 4, num_residual=1)
            self.img_net4 = GET_IMAGE_G(self.gf_dim // 8)
        if n > 4:
            self.h_net4 = NEXT_STAGE_G(self.gf_dim // 8, num_residual=1)
            self.img_net4 = GET_IMAGE_G(self.gf_dim // 16)

    def forward(self, z_code, text_embedding=None, eps=None, nhwc=False):
        """z_code (B, Z_DIM), text_embedding (B, TEXT.DIMENSION) -> ([images NCHW], mu, logvar).
        `eps` optionally pins the reparameterisation noise (parity tests); default: torch's RNG.
        `nhwc=True` (evaluation path) returns the images in the kernels' own NHWC4 layout."""
        if self.b_condition and text_embedding is not None:
            c_code, mu, logvar = self.ca_net(text_embedding, eps)
        else:
            c_code, mu, logvar = z_code, None, None
        fake_imgs = []
        h = None
        for i in range(min(self.branch_num, 4)):
            h_net = getattr(self, 'h_net%d' % (i + 1))
            h = h_net(z_code, c_code) if i == 0 else h_net(h, c_code)
            fake_imgs.append(getattr(self, 'img_net%d' % (i + 1))(h, nhwc))
        return fake_imgs, mu, logvar


# ---- discriminators -------------------------------------------------------------------------------------------
# tower beyond the shared /16 encoder: (attribute name, block, in multiple of ndf, out multiple of ndf)
_TOWERS = {
    64: (),
    128: (("img_code_s32", "down", 8, 16), ("img_code_s32_1", "same", 16, 8)),
    256: (("img_code_s32", "down", 8, 16), ("img_code_s64", "down", 16, 32),
          ("img_code_s64_1", "same", 32, 16), ("img_code_s64_2", "same", 16, 8)),
    512: (("img_code_s32", "down", 8, 16), ("img_code_s64", "down", 16, 32), ("img_code_s128", "down", 32, 64),
          ("img_code_s128_1", "same", 64, 32), ("img_code_s128_2", "same", 32, 16), ("img_code_s128_3", "same", 16, 8)),
    1024: (("img_code_s32", "down", 8, 16), ("img_code_s64", "down", 16, 32), ("img_code_s128", "down", 32, 64),
           ("img_code_s256", "down", 64, 128), ("img_code_s256_1", "same", 128, 64), ("img_code_s256_2", "same", 64, 32),
           ("img_code_s256_3", "same", 32, 16), ("img_code_s256_4", "same", 16, 8)),
}


class _DNet(nn.Module):
    """Per-scale discriminator: stride-2 tower to a 4x4 map, conditional head (jointConv with c_code
    + logits) and unconditional head (model.py:402-551).  Attribute order = the reference's
    definition order, which fixes state_dict order and RNG consumption."""
    size = 64

    def __init__(self):
        super().__init__()
        ndf = self.df_dim = cfg.GAN.DF_DIM
        self.ef_dim = cfg.GAN.EMBEDDING_DIM
        self.b_condition = cfg.GAN.B_CONDITION
        self.img_code_s16 = encode_image_by_16times(ndf)
        self._tower = []
        for name, block, cin, cout in _TOWERS[self.size]:
            setattr(self, name, downBlock(ndf * cin, ndf * cout) if block == "down"
                    else Block3x3_leakRelu(ndf * cin, ndf * cout))
            self._tower.append(name)
        self.logits = nn.Sequential(nn.Conv2d(ndf * 8, 1, kernel_size=4, stride=4), nn.Sigmoid())
        if self.b_condition:
            self.jointConv = Block3x3_leakRelu(ndf * 8 + self.ef_dim, ndf * 8)
            self.uncond_logits = nn.Sequential(nn.Conv2d(ndf * 8, 1, kernel_size=4, stride=4), nn.Sigmoid())

    def forward(self, x_var, c_code=None, groups=1, need_features=True, taps=None):
        """`groups` > 1 (not part of the reference signature): x_var stacks that many independent batches
        along dim 0 — the real / wrong / fake passes of trainer.py:390-392 in one launch per layer — and
        every BatchNorm keeps separate statistics per batch, updating its running statistics in that order.
        `taps` (a list, tests only) receives the NHWC output of every LeakyReLU block in forward order."""
        x = ops.ToNHWC.apply(x_var, 4)
        x_code = self.img_code_s16(x, groups, taps, defer_last=bool(self._tower))
        hook = getattr(self, 'after_s16_hook', None)
        if hook is not None and x_code.requires_grad:
            # fires when the gradient of img_code_s16's output exists, i.e. when the parameter gradients of the tower and
            # the heads (96 % of D_NET256's parameters) have been issued: the data-parallel trainer starts reducing that
            # part of the flat gradient there, under the backward of the four image-side convolutions
            x_code.register_hook(hook)
        for k, name in enumerate(self._tower):
            # every tower block but the last feeds only the next block (the last one's map goes to three consumers)
            x_code = getattr(self, name)(x_code, groups=groups, defer=taps is None and k + 1 < len(self._tower))
            if taps is not None:
                taps.append(x_code)
        B, C = x_code.shape[0], x_code.shape[3]
        # the reference flattens an NCHW map (model.py:428): keep that element order for callers
        x_immediate = ops.ToNCHW.apply(x_code, C).reshape(B, -1) if need_features else None
        if self.b_condition and c_code is not None:
            h_c_code = self.jointConv(x_code, cvec=c_code.reshape(-1, self.ef_dim), groups=groups)
            if taps is not None:
                taps.append(h_c_code)
        else:
            h_c_code = x_code
        output = ops.LogitHead.apply(h_c_code, self.logits[0].weight, self.logits[0].bias)
        if self.b_condition:
            out_uncond = ops.LogitHead.apply(x_code, self.uncond_logits[0].weight, self.uncond_logits[0].bias)
            return [output.view(-1), out_uncond.view(-1)], x_immediate
        return [output.view(-1)], x_immediate


class D_NET64(_DNet):
    size = 64


class D_NET128(_DNet):
    size = 128


class D_NET256(_DNet):
    size = 256


class D_NET512(_DNet):
    size = 512


class D_NET1024(_DNet):
    size = 1024


class INCEPTION_V3(nn.Module):
    """Placeholder for the reference's Inception-v3 scorer (model.py:17-109), which downloads weights
    at construction and needs torchvision: evaluation metrics are outside the train-step path
    (SURVEY.md §2).  Importable so `from model import ... INCEPTION_V3` works; not callable."""

    def __init__(self):
        super().__init__()

    def forward(self, x):
        raise RuntimeError("INCEPTION_V3 (IS/FID scoring) is out of scope of the MI355X train-step path")
