"""bf16 activation mode (BASELINE config 4: bf16 activations / weights in HBM, bf16 MFMA, fp32 accumulate, fp32
BatchNorm statistics, fp32 master weights and Adam) on a real MI355X.

BASELINE.json states no tolerance for bf16 ("report max-abs / rel error vs the fp32 path").  Bounds
used here, and why:
  * a convolution alone, against torch fp32 on the SAME bf16-rounded operands: products of bf16 numbers are exact in
    fp32 and the accumulation is fp32, so the only error is the final rounding of the result to bf16: 2^-8 relative
    (+ a small absolute floor for cancelling sums); the BatchNorm partial sums come from the fp32 accumulators and are
    held to 1e-4;
  * a fused block and whole networks: bf16 carries 8 significant bits, every stored activation rounds once, so
    deviations of ~1e-2 in relative L2 norm are the expected size; the tests print what they measure.
"""
import zlib

import pytest
import torch
import torch.nn.functional as F

from helpers import CASES, build_nets, make_batch

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def r16(t):
    """Round to the nearest bf16 value, keep fp32 storage (what the GPU path sees)."""
    return t.to(torch.bfloat16).float()


def rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture
def bf16_mode():
    from speech_to_image_translation_without_text_amd import ops
    old = ops.ACT_BF16
    ops.ACT_BF16 = True
    yield
    ops.ACT_BF16 = old


def ref_conv(x, w, kind):
    if kind == "up":
        return F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    if kind == "k3s1":
        return F.conv2d(x, w, padding=1)
    return F.conv2d(x, w, stride=2, padding=1)


CONV_CASES = [
    # kind, B, H, Cin, Cout            tile / (BN, CK) class it exercises
    ("k3s1", 2, 64, 64, 128),         # 4x32 tile, (128,16)
    ("k3s1", 3, 16, 128, 64),         # 8x16 tile, (64,32), 4 chunks
    ("k3s1", 5, 4, 256, 512),         # 8 whole 4x4 maps per tile (ragged: 5 images), split-K
    ("k3s1", 2, 32, 32, 32),          # (32,32)
    ("k3s1", 2, 32, 64, 32),          # (32,64)
    ("k3s1", 2, 8, 64, 64),           # 2 images of 8x8 per tile
    ("k4s2", 2, 64, 64, 128),         # stride 2: de-interleaved patch columns, two tap groups
    ("k4s2", 3, 8, 128, 256),         # 4x4 outputs: 8 maps per tile, 800-pixel patch
    ("k4s2", 2, 32, 32, 64),          # (64,32)
    ("k4s2", 2, 64, 32, 32),          # (32,32)
    ("k4s2", 9, 16, 64, 128),         # 8x8 outputs, 2 maps per tile, ragged batch
    ("up", 2, 32, 64, 128),           # 4 phases, (128,32)
    ("up", 9, 4, 256, 256),           # 8 maps of 4x4 per tile, ragged batch, split-K
    ("up", 2, 64, 64, 64),            # (64,64)
    ("up", 2, 64, 32, 32),            # (32,32)
    ("up", 2, 32, 64, 32),            # (32,64)
]


# cases for the 256-pixel second-generation kernel (tuning knobs b16_v2=2: wherever it can run; b16_persist=4: four block
# slots, so that a block walks several tiles)
V2_CASES = [
    ("k4s2", 3, 64, 64, 128),         # 12 tiles of 8x32 outputs over 4 persistent blocks: top / bottom halo rows, 2 chunks
    ("k4s2", 2, 128, 32, 128),        # two tiles across a row (left / right halo columns), ONE channel chunk
    ("k4s2", 5, 32, 128, 256),        # one 16x16 map per tile, two channel blocks, 4 chunks, odd tile count
    ("k4s2", 2, 64, 64, 128),
    ("k3s1", 2, 64, 64, 128),         # two patch buffers, 3 stages per chunk (register sets swap roles per chunk)
    ("k3s1", 5, 8, 96, 256),          # 4 images per tile, ragged batch, 3 chunks (odd)
    ("k3s1", 3, 16, 32, 192),         # one chunk; 192 channels = 1.5 channel blocks
    ("up", 2, 32, 64, 128),           # transposed phases: halo sides depend on the phase parity; one chunk of 64
    ("up", 3, 16, 128, 128),          # 2 chunks
    ("up", 5, 8, 192, 256),           # 4 images per tile, ragged, 3 chunks
]


@pytest.mark.parametrize("case", V2_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_bf16_conv_second_generation_kernels(gpu, case):
    """The same check through conv_bf16_v2_kernel, forced wherever it is eligible, with few enough block slots that the
    persistent form walks several tiles per block."""
    from speech_to_image_translation_without_text_amd import _lib
    with _lib.tuning(b16_v2=2, b16_persist=4):
        _conv_case(gpu, case)


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_bf16_conv_kernels_against_fp32_on_rounded_operands(gpu, case):
    """Forward, input gradient and weight gradient of one convolution through the bf16 kernels."""
    _conv_case(gpu, case)


@pytest.mark.parametrize("bm", [128, 256, 512])
@pytest.mark.parametrize("case", [("k4s2", 2, 64, 64, 128), ("k3s1", 5, 4, 256, 512), ("k4s2", 9, 16, 64, 128),
                                  ("up", 2, 32, 64, 128), ("k4s2", 3, 32, 128, 256)], ids=lambda c: "-".join(str(v) for v in c))
def test_bf16_weight_gradient_tile_heights(gpu, case, bm):
    """igemm_wgrad_b16_kernel with 128 x 128 and with 256 x 128 tiles (wgrad16_bm forces either wherever 256 divides the taps x
    channels rows; the planner otherwise picks by its cost model): same reference, same bounds."""
    from speech_to_image_translation_without_text_amd import _lib
    with _lib.tuning(wgrad16_bm=bm):
        _conv_case(gpu, case)


def _conv_case(gpu, case):
    from speech_to_image_translation_without_text_amd import ops
    from speech_to_image_translation_without_text_amd._lib import PACK_PLAIN, PACK_UPFOLD
    kind, B, H, Cin, Cout = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 100000)
    kk = 4 if kind == "k4s2" else 3
    x = r16(torch.randn(B, Cin, H, H, generator=g))
    w = r16(torch.randn(Cout, Cin, kk, kk, generator=g) / (kk * Cin ** 0.5))
    Ho = {"k3s1": H, "k4s2": H // 2, "up": 2 * H}[kind]
    gy = r16(torch.randn(B, Cout, Ho, Ho, generator=g))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yref = ref_conv(xr, wr, kind)
    yref.backward(gy)

    wg = w.to(gpu)
    packed = ops.pack_weight(wg, PACK_UPFOLD if kind == "up" else PACK_PLAIN)
    if kind == "up":
        # the folded 4x4 taps are sums of up to four bf16 weights: compare against the reference on THOSE rounded values
        pass
    xg = nhwc(x).to(gpu).to(torch.bfloat16)
    gyg = nhwc(gy).to(gpu).to(torch.bfloat16)
    y, part, nparts = ops.conv_any(ops._KIND[kind], xg, packed, Cout, stats=True, out_dtype=torch.bfloat16)
    dx = ops._dgrad(kind, gyg, wg, packed, Cin, out_dtype=torch.bfloat16)
    wparam = torch.nn.Parameter(wg.clone())
    dw = ops._wgrad(kind, xg, None, gyg, wparam)
    torch.cuda.synchronize()
    assert y.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16 and dw.dtype == torch.float32

    def check(got, ref, what, rel=2.0 ** -7, floor=2e-3):
        got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
        assert got.shape == ref.shape, (what, got.shape, ref.shape)
        scale = float(ref.abs().max())
        err = (got - ref).abs()
        bad = err > rel * ref.abs() + floor * scale
        assert not bool(bad.any()), "%s: max err %.3e (scale %.3e), %d/%d beyond" % (what, float(err.max()), scale,
                                                                                    int(bad.sum()), bad.numel())
    # "up": the 3x3 weights are folded to 4x4 sums in fp32 and then rounded to bf16 once more (<= 2^-9 per tap)
    extra = 3.0 if kind == "up" else 1.0
    check(nchw(y.float()), yref, "y", rel=extra * 2.0 ** -7)
    check(nchw(dx.float()), xr.grad, "dx", rel=extra * 2.0 ** -7)
    check(dw, wr.grad, "dw", rel=1e-3, floor=1e-3)   # fp32 result of exact products: only the summation order differs
    # BatchNorm partial sums come from the fp32 accumulators, before the rounding of y
    s1 = part[0].sum(0).cpu().double()
    s2 = part[1].sum(0).cpu().double()
    yd = yref.detach().double()
    ref1, ref2 = yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))
    tol = (3e-3 if kind == "up" else 2e-4)
    assert float((s1 - ref1).abs().max()) <= tol * float(yd.abs().sum((0, 2, 3)).max()), "column sums"
    assert float((s2 - ref2).abs().max()) <= 2 * tol * float(ref2.max()), "column sums of squares"


BLOCK_CASES = [
    # kind, B, H, Cx, Cc, Cout, act, residual
    ("k3s1", 2, 32, 64, 128, 128, "glu", False),   # G jointConv: c_code folded into a class bias
    ("k3s1", 2, 32, 64, 0, 128, "glu", False),     # ResBlock first half
    ("k3s1", 2, 32, 64, 0, 64, "none", True),      # ResBlock second half (residual add)
    ("k3s1", 8, 4, 128, 32, 64, "lrelu", False),   # D jointConv on 4x4 maps: c_code concat materialised
    ("k4s2", 4, 32, 64, 0, 128, "lrelu", False),
    ("k4s2", 8, 8, 256, 0, 512, "lrelu", False),   # split-K, statistics in the slab reduction
    ("up", 4, 4, 512, 0, 512, "glu", False),
    ("up", 2, 32, 64, 0, 64, "glu", False),
    ("k3s1", 2, 16, 16, 0, 32, "glu", False),      # not eligible for the bf16 kernel (16 channels): fp32 MFMA, bf16 storage
    ("k4s2", 2, 16, 8, 0, 16, "lrelu", False),     # ... 8 channels
]


@pytest.mark.parametrize("case", [BLOCK_CASES[0], BLOCK_CASES[1], BLOCK_CASES[4]],
                         ids=lambda c: "-".join(str(v) for v in c))
def test_bf16_fused_block_second_generation_kernels(gpu, case, bf16_mode):
    """The fused block (class-bias epilogue, BatchNorm sums per tile, both gradients) through conv_bf16_v2_kernel: the three
    block cases with more than 64 output channels."""
    from speech_to_image_translation_without_text_amd import _lib
    with _lib.tuning(b16_v2=2, b16_persist=4):
        _block_case(gpu, case)


@pytest.mark.parametrize("case", BLOCK_CASES, ids=lambda c: "-".join(str(v) for v in c))
def test_bf16_fused_block_tracks_fp32_reference(gpu, case, bf16_mode):
    _block_case(gpu, case)


def _block_case(gpu, case):
    """conv + BatchNorm + GLU / LeakyReLU / residual with bf16 storage, forward and backward, against torch fp32 on the
    same (bf16-rounded) inputs.  Relative L2 deviation is printed and bounded by 1.5e-2 (outputs) / 1e-2 (gradients; 6e-2
    behind a LeakyReLU, see below)."""
    from test_kernels_gpu import ACT, ref_block
    from speech_to_image_translation_without_text_amd import ops
    kind, B, H, Cx, Cc, Cout, act, use_res = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 100000)
    kk = {"k3s1": 3, "k4s2": 4, "up": 3}[kind]
    x = r16(torch.randn(B, Cx, H, H, generator=g))
    cvec = torch.randn(B, Cc, generator=g) if Cc else None
    w = torch.randn(Cout, Cx + Cc, kk, kk, generator=g) * (1.0 / (kk * (Cx + Cc) ** 0.5))
    gamma = 1 + 0.1 * torch.randn(Cout, generator=g)
    beta = 0.1 * torch.randn(Cout, generator=g)
    Ho = {"k3s1": H, "k4s2": H // 2, "up": 2 * H}[kind]
    Cact = Cout // 2 if act == "glu" else Cout
    res = r16(torch.randn(B, Cact, Ho, Ho, generator=g)) if use_res else None
    gout = r16(torch.randn(B, Cact, Ho, Ho, generator=g))
    leaves = [t.clone().requires_grad_(True) if t is not None else None for t in (x, cvec, w, gamma, beta, res)]
    ref = ref_block(*leaves, kind, act)
    ref.backward(gout)

    dl = [t.clone().to(gpu).requires_grad_(True) if t is not None else None for t in (x, cvec, w, gamma, beta, res)]
    xg = nhwc(dl[0].detach()).to(torch.bfloat16).requires_grad_(True)
    resg = nhwc(dl[5].detach()).to(torch.bfloat16).requires_grad_(True) if use_res else None
    rm, rv = torch.zeros(Cout, device=gpu), torch.ones(Cout, device=gpu)
    nbt = torch.zeros((), dtype=torch.long, device=gpu)
    out = ops.ConvBnAct.apply(xg, dl[1], dl[2], dl[3], dl[4], resg, kind, ACT[act], (rm, rv, nbt), True)
    assert out.dtype == torch.bfloat16
    out.backward(nhwc(gout.to(gpu)).to(torch.bfloat16))
    torch.cuda.synchronize()
    dev = dict(out=rel_l2(nchw(out.float()), ref), dx=rel_l2(nchw(xg.grad.float()), leaves[0].grad),
               dw=rel_l2(dl[2].grad, leaves[2].grad), dgamma=rel_l2(dl[3].grad, leaves[3].grad),
               dbeta=rel_l2(dl[4].grad, leaves[4].grad))
    if Cc:
        dev['dcvec'] = rel_l2(dl[1].grad, leaves[1].grad)
    if use_res:
        dev['dres'] = rel_l2(nchw(resg.grad.float()), leaves[5].grad)
    print("bf16 block %s: relative L2 deviation from fp32 " % (case,), {k: "%.2e" % v for k, v in dev.items()})
    # LeakyReLU blocks: the sign of scale * y + shift is taken from the bf16-rounded y, so pre-activations within 2^-9 of
    # zero flip their slope (1 vs 0.2) against the fp32 reference: ~0.1 % of the elements, a few 1e-2 in the gradients'
    # norm (the forward value of such an element is ~0 either way)
    gtol = 6e-2 if act == "lrelu" else 1e-2
    for k, v in dev.items():
        assert v <= (1.5e-2 if k == "out" else gtol), (k, v, dev)
    # running statistics: fp32 accumulators, so nearly the fp32 values
    with torch.no_grad():
        xin = leaves[0] if cvec is None else torch.cat((leaves[1].view(B, -1, 1, 1).repeat(1, 1, H, H), leaves[0]), 1)
        yraw = ref_block(xin, None, leaves[2], None, None, None, kind, "none")
        m = yraw.transpose(0, 1).reshape(Cout, -1)
    assert rel_l2(rm, 0.1 * m.mean(1)) <= 2e-2 and rel_l2(rv, 0.9 + 0.1 * m.var(1, unbiased=True)) <= 1e-3
    assert int(nbt.item()) == 1


IMAGE_LAYERS = [
    # (kind, B, H, Cin true, Cin stored, Cout true, n_out, act): the 3-channel edges of the bf16 mode
    ("k3s1", 2, 64, 16, 16, 3, 4, "tanh"),     # GET_IMAGE_G: bf16 features -> fp32 RGB (pixels-as-columns MFMA kernel)
    ("k3s1", 2, 64, 32, 32, 3, 4, "tanh"),
    ("k3s1", 3, 64, 64, 64, 3, 4, "tanh"),
    ("k4s2", 2, 64, 3, 4, 64, 64, "lrelu"),    # first discriminator conv: fp32 NHWC4 image -> bf16 features
    ("k4s2", 2, 128, 3, 4, 32, 32, "lrelu"),
    ("k4s2", 3, 64, 3, 4, 16, 16, "lrelu"),
]


@pytest.mark.parametrize("case", IMAGE_LAYERS, ids=lambda c: "-".join(str(v) for v in c))
def test_bf16_image_layers_forward_and_gradients(gpu, case, bf16_mode):
    """ConvAct at the image edges of the bf16 mode (model.py:287-298, 383-384): forward, input gradient and weight gradient
    against torch fp32 on the same inputs; bf16 products with fp32 accumulation: 1.5e-2 relative L2."""
    from test_kernels_gpu import ACT
    from speech_to_image_translation_without_text_amd import ops
    kind, B, H, Cin, Cs, Cout, n_out, act = case
    g = torch.Generator().manual_seed(zlib.crc32(repr(case).encode()) % 100000)
    kk = 3 if kind == "k3s1" else 4
    x = r16(torch.randn(B, Cin, H, H, generator=g))
    w = torch.randn(Cout, Cin, kk, kk, generator=g) / (kk * Cin ** 0.5)
    Ho = H if kind == "k3s1" else H // 2
    gout = r16(torch.randn(B, Cout, Ho, Ho, generator=g))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, padding=1, stride=1 if kind == "k3s1" else 2)
    y = torch.tanh(y) if act == "tanh" else F.leaky_relu(y, 0.2)
    y.backward(gout)
    xs = torch.zeros(B, Cs, H, H)
    xs[:, :Cin] = x
    xg = nhwc(xs).to(gpu)
    xg = (xg.to(torch.bfloat16) if Cs >= 8 else xg).requires_grad_(True)
    wg = w.to(gpu).requires_grad_(True)
    out = ops.ConvAct.apply(xg, wg, None, kind, ACT[act], n_out)
    assert out.dtype == (torch.bfloat16 if n_out >= 8 else torch.float32)
    gs = torch.zeros(B, n_out, Ho, Ho)
    gs[:, :Cout] = gout
    out.backward(nhwc(gs).to(gpu).to(out.dtype))
    torch.cuda.synchronize()
    dev = dict(out=rel_l2(nchw(out.float())[:, :Cout], y), dx=rel_l2(nchw(xg.grad.float())[:, :Cin], xr.grad),
               dw=rel_l2(wg.grad, wr.grad))
    print("bf16 image layer %s: relative L2 deviation from fp32" % (case,), {k: "%.2e" % v for k, v in dev.items()})
    for k, v in dev.items():
        assert v <= (6e-2 if (act == "lrelu" and k != "out") else 1.5e-2), (k, v, dev)


def _run_steps(gpu, case, n_steps, bf16):
    from speech_to_image_translation_without_text_amd import ops, trainer as T
    old = ops.ACT_BF16
    ops.ACT_BF16 = bf16
    try:
        netG, netsD = build_nets(case)
        batch = make_batch(case)
        netG.to(gpu)
        for d in netsD:
            d.to(gpu)
        tr = T.condGANTrainer(None, None, 256, False)
        tr.build(netG, netsD)
        tr.flatG.lr = 0.0 if n_steps == 1 else tr.flatG.lr
        b = {k: ([t.to(gpu) for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else
                 (v.to(gpu) if torch.is_tensor(v) else v)) for k, v in batch.items()}
        gen = torch.Generator(device=gpu).manual_seed(77)
        losses = []
        for it in range(n_steps):
            noise = b['noise'] if it == 0 else torch.randn(b['noise'].shape, device=gpu, generator=gen)
            eps = b['eps'] if it == 0 else torch.randn(b['eps'].shape, device=gpu, generator=gen)
            emb = b['emb'].clone().requires_grad_(True)
            out = tr.train_step(b['real'], b['wrong'], emb, batch['labels'], noise, eps)
            losses.append([float(o) for o in out])
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in netG.named_parameters()}
        return dict(losses=losses, fakes=[f.detach().float().clone() for f in tr.fake_imgs], grads=grads,
                    grad_emb=emb.grad.detach().clone())
    finally:
        ops.ACT_BF16 = old


def test_bf16_step_small_net_tracks_fp32(gpu):
    """One full iteration of the reduced-width three-stage nets (every block type, mostly the fp32-MFMA / bf16-storage
    fallback because the channel counts are below 32) in bf16 mode against the fp32 HIP path."""
    case = dict(CASES['small3'], B=8)
    a = _run_steps(gpu, case, 1, False)
    c = _run_steps(gpu, case, 1, True)
    print("small3 bf16 vs fp32 losses", a['losses'], c['losses'])
    for la, lc in zip(a['losses'][0], c['losses'][0]):
        assert abs(la - lc) <= 0.01 * abs(la) + 2e-3, (a['losses'], c['losses'])
    for i in range(3):
        r = rel_l2(c['fakes'][i], a['fakes'][i])
        print("small3 img%d: rel L2 %.3e, max abs %.3e" % (i, r, float((c['fakes'][i] - a['fakes'][i]).abs().max())))
        assert r < 0.05


def test_bf16_config4_full_width_batch48_tracks_fp32(gpu):
    """BASELINE config 4 at its workload: branch_num=3, full width, batch 48, bf16 activations.  One full iteration
    (G forward, three D updates, G update with lr_G = 0 so that G's gradients stay readable) in bf16 mode against the
    fp32 HIP path on the same seeded weights and inputs.  Reports max-abs and relative-L2 deviation of the images, the
    losses and G's gradients and bounds them at about 1.5 - 2x what is measured (round 3: images
    8.4e-3 rel L2 / 3.7e-2 max abs, losses 0.1 %, end-to-end G gradients 0.33, segments G 2.2e-2, D 8.5e-2 / 1.2e-1 / 1.5e-1): images
    1.5e-2 / 8e-2, losses 1 %, end to end 0.45, segments 3e-2 and 1.2e-1 / 1.6e-1 / 2e-1."""
    case = dict(CASES['full3_fwd'], B=48)
    a = _run_steps(gpu, case, 1, False)
    torch.cuda.empty_cache()
    c = _run_steps(gpu, case, 1, True)
    print("config 4 (B=48) losses fp32 %s | bf16 %s" % (a['losses'][0], c['losses'][0]))
    for la, lc in zip(a['losses'][0], c['losses'][0]):
        assert abs(la - lc) <= 0.01 * abs(la) + 2e-3, (a['losses'], c['losses'])
    for i in range(3):
        r = rel_l2(c['fakes'][i], a['fakes'][i])
        mx = float((c['fakes'][i] - a['fakes'][i]).abs().max())
        print("config 4 img%d (%dpx): rel L2 %.3e, max abs %.3e" % (i, 64 << i, r, mx))
        assert r < 1.5e-2 and mx < 8e-2, (i, r, mx)
    # end to end G's gradients pass three discriminators that each side updated itself (first-step Adam = lr * sign(g))
    # and their LeakyReLU chains: reported, loosely bounded; the segments below carry the real bounds
    worst = ("", 0.0)
    for k, gref in a['grads'].items():
        r = rel_l2(c['grads'][k], gref)
        if gref.numel() >= 4096 and r > worst[1]:
            worst = (k, r)
    print("config 4 G gradients END TO END: worst rel L2 deviation over tensors >= 4096 elements: %s %.3e; grad_emb %.3e"
          % (worst[0], worst[1], rel_l2(c['grad_emb'], a['grad_emb'])))
    assert worst[1] < 0.45, worst
    del a, c
    torch.cuda.empty_cache()
    sa = _segment_grads(gpu, case, False)
    torch.cuda.empty_cache()
    sc = _segment_grads(gpu, case, True)
    # D: every LeakyReLU decides its slope from a bf16-rounded pre-activation, ~0.1 % of the decisions differ from the fp32
    # path per layer (2.5e-2 of the gradient norm per layer, test_bf16_fused_block...), up to eight layers deep
    for seg, tol in (("G", 3e-2), ("D0", 1.2e-1), ("D1", 1.6e-1), ("D2", 2e-1)):
        worst = ("", 0.0)
        for k, gref in sa[seg].items():
            r = rel_l2(sc[seg][k], gref)
            if gref.numel() >= 4096 and r > worst[1]:
                worst = (k, r)
        print("config 4 segment %s (identical weights, inputs and cotangents): worst rel L2 deviation of a parameter "
              "gradient (>= 4096 elements): %s %.3e" % (seg, worst[0], worst[1]))
        assert worst[1] < tol, (seg, worst)


def test_bf16_config4_twenty_iterations_track_fp32(gpu):
    """Config 4's workload followed for 20 iterations (full width, batch 48, fresh noise per step, both networks training):
    the bf16 run's losses against the fp32 HIP run's on the same seeds.  GAN training amplifies differences, so the band is on
    the trajectory: errD / errG within 15 % (+ 0.05 absolute) of the fp32 value at every iteration, KL within 5 %, nothing
    non-finite (reported: the largest deviations)."""
    case = dict(CASES['full3_fwd'], B=48)
    a = _run_steps(gpu, case, 20, False)
    torch.cuda.empty_cache()
    c = _run_steps(gpu, case, 20, True)
    worst = [0.0, 0.0, 0.0]
    for it, (la, lc) in enumerate(zip(a['losses'], c['losses'])):
        assert all(v == v and abs(v) < 1e6 for v in lc), (it, lc)
        for k in range(3):
            worst[k] = max(worst[k], abs(la[k] - lc[k]) / (abs(la[k]) + 1e-9))
    print("config 4, 20 iterations: fp32 first / last %s / %s; bf16 first / last %s / %s; worst relative deviation errD %.3f errG %.3f kl %.3f"
          % (a['losses'][0], a['losses'][-1], c['losses'][0], c['losses'][-1], worst[0], worst[1], worst[2]))
    for it, (la, lc) in enumerate(zip(a['losses'], c['losses'])):
        assert abs(la[0] - lc[0]) <= 0.15 * abs(la[0]) + 0.05, ("errD", it, la, lc)
        assert abs(la[1] - lc[1]) <= 0.15 * abs(la[1]) + 0.05, ("errG", it, la, lc)
        assert abs(la[2] - lc[2]) <= 0.05 * abs(la[2]) + 1e-3, ("kl", it, la, lc)


def _segment_grads(gpu, case, bf16):
    """Parameter gradients of G alone (fixed cotangents on the three images) and of each D alone (fixed images, the
    six-term D loss), from identical seeded weights: what bf16 storage changes in ONE network's forward + backward."""
    import torch.nn as nn
    from speech_to_image_translation_without_text_amd import ops
    old = ops.ACT_BF16
    ops.ACT_BF16 = bf16
    try:
        netG, netsD = build_nets(case)
        batch = make_batch(case)
        B = case['B']
        netG.to(gpu)
        gen = torch.Generator(device=gpu).manual_seed(99)
        emb = batch['emb'].to(gpu).requires_grad_(True)
        fakes, mu, logvar = netG(batch['noise'].to(gpu), emb, batch['eps'].to(gpu))
        cots = [torch.randn(f.shape, device=gpu, generator=gen) / f[0].numel() for f in fakes]
        torch.autograd.backward(list(fakes), cots)
        out = {"G": {k: p.grad.detach().clone() for k, p in netG.named_parameters() if p.grad is not None}}
        crit = nn.BCELoss()
        ones, zeros = torch.ones(B, device=gpu), torch.zeros(B, device=gpu)
        c = torch.randn(B, case['ef'], device=gpu, generator=gen)
        for i, d in enumerate(netsD):
            d.to(gpu)
            loss = 0
            for tc, tu in ((ones, ones), (zeros, ones), (zeros, zeros)):
                img = torch.rand(B, 3, 64 << i, 64 << i, device=gpu, generator=gen) * 2 - 1
                logits, _ = d(img, c)
                loss = loss + crit(logits[0], tc) + crit(logits[1], tu)
            loss.backward()
            out["D%d" % i] = {k: p.grad.detach().clone() for k, p in d.named_parameters()}
            d.cpu()
        torch.cuda.synchronize()
        return out
    finally:
        ops.ACT_BF16 = old


def test_bf16_training_stays_finite_and_close_over_iterations(gpu):
    """Five iterations at reduced width: losses of the bf16 path stay finite and follow the fp32 path's trajectory."""
    case = dict(CASES['small3'], B=8)
    a = _run_steps(gpu, case, 5, False)
    c = _run_steps(gpu, case, 5, True)
    print("5 iterations: fp32", a['losses'][-1], "bf16", c['losses'][-1])
    for la, lc in zip(a['losses'], c['losses']):
        for u, v in zip(la, lc):
            assert v == v and abs(v) < 1e4
            assert abs(u - v) <= 0.15 * abs(u) + 5e-2, (a['losses'], c['losses'])
