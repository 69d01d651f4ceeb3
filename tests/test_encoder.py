"""Speech-encoder front-end (BASELINE config 5): seeded construction and the oracle
against the reference's outputs on the CPU; the HIP path against both on the GPU."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_close, checksum

_spec = importlib.util.spec_from_file_location("make_golden_encoder", os.path.join(GOLDEN, "make_golden_encoder.py"))
mge = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mge)


def build_encoder():
    from speech_to_image_translation_without_text_amd.speech_encoder import CNNRNN
    torch.manual_seed(0)
    net = CNNRNN(40, embedding_dim=1024, nhidden=1024, nsent=1024, bidirectional=True, rnn_layers=1)
    g = torch.Generator().manual_seed(5)
    for k, v in net.state_dict().items():
        if k.endswith('running_mean'):
            v.copy_(0.2 * torch.randn(v.shape, generator=g))
        elif k.endswith('running_var'):
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
    return net.eval()


def test_encoder_construction_and_oracle_match_reference():
    from oracle import speech_encoder_oracle as orc
    gold = np.load(os.path.join(GOLDEN, "encoder.npz"), allow_pickle=False)
    net = build_encoder()
    assert list(net.state_dict().keys()) == [str(k) for k in gold['keys']]
    np.testing.assert_allclose(checksum(net.state_dict()), gold['checksum'], rtol=0, atol=0)
    x, lens = mge.make_inputs()
    assert lens.tolist() == gold['lens'].tolist()
    with torch.no_grad():
        words, sent = orc.forward({k: v.clone() for k, v in net.state_dict().items()}, x, lens, 512, True)
    assert_close(words, gold['words'], rtol=1e-3, atol=1e-5, what="words_emb")
    assert_close(sent, gold['sent'], rtol=1e-3, atol=1e-6, what="sent_emb")
    # padded steps of the shorter sequences are zero, and the mean runs over all 32 steps (speech_encoder.py:88-93)
    assert float(words[1, :, int(lens[1]):].abs().max()) == 0.0


def test_encoder_training_mode_is_refused():
    from speech_to_image_translation_without_text_amd import _lib
    net = build_encoder().train()
    with pytest.raises(_lib.S2IError):
        net(torch.zeros(2, 40, 2048), torch.tensor([32, 32]))


@pytest.mark.gpu
def test_encoder_hip_matches_reference(gpu):
    gold = np.load(os.path.join(GOLDEN, "encoder.npz"), allow_pickle=False)
    net = build_encoder().to(gpu)
    x, lens = mge.make_inputs()
    words, sent = net(x.to(gpu), lens)
    torch.cuda.synchronize()
    assert words.shape == (3, 1024, 32) and sent.shape == (3, 1024)
    assert_close(words, gold['words'], rtol=1e-3, atol=1e-4, what="words_emb")
    assert_close(sent, gold['sent'], rtol=1e-3, atol=1e-5, what="sent_emb")
    assert float(words[2, :, int(lens[2]):].abs().max()) == 0.0
    # feeds the generator: (B, 1024) embedding
    assert net.extract_feature(x.to(gpu), lens).shape == (3, 1024)


def _small_encoder(bidirectional, nhidden, seed=3):
    from speech_to_image_translation_without_text_amd.speech_encoder import CNNRNN
    torch.manual_seed(seed)
    net = CNNRNN(40, embedding_dim=1024, nhidden=nhidden, nsent=nhidden, bidirectional=bidirectional, rnn_layers=1)
    g = torch.Generator().manual_seed(5)
    for k, v in net.state_dict().items():
        if k.endswith('running_mean'):
            v.copy_(0.2 * torch.randn(v.shape, generator=g))
        elif k.endswith('running_var'):
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
    return net.eval()


@pytest.mark.gpu
@pytest.mark.parametrize("B,bidirectional,nhidden", [(33, True, 64), (4, False, 1024), (4, True, 64)])
def test_encoder_fallback_branches_against_oracle(gpu, B, bidirectional, nhidden):
    """The branches the B = 3 golden does not reach (Audio_to_Image/speech_encoder.py:40-42, 84-90): more than 32
    sequences, a unidirectional LSTM and a hidden size above 512 take the per-direction recurrent-GEMM path; the
    small bidirectional case takes the fused step kernel.  Checked against the oracle (pinned by encoder.npz)."""
    from oracle import speech_encoder_oracle as orc
    net = _small_encoder(bidirectional, nhidden)
    g = torch.Generator().manual_seed(9)
    T = 512                                         # 8 LSTM steps
    x = torch.randn(B, 40, T, generator=g) * 20 - 40
    lens = torch.sort(torch.randint(1, T // 64 + 1, (B,), generator=g), descending=True)[0]
    hd = nhidden // 2 if bidirectional else nhidden
    with torch.no_grad():
        words_o, sent_o = orc.forward({k: v.clone() for k, v in net.state_dict().items()}, x, lens, hd, bidirectional)
    net.to(gpu)
    words, sent = net(x.to(gpu), lens)
    torch.cuda.synchronize()
    assert_close(words, words_o, rtol=1e-3, atol=1e-4, what="words_emb")
    assert_close(sent, sent_o, rtol=1e-3, atol=1e-5, what="sent_emb")


@pytest.mark.gpu
@pytest.mark.parametrize("width", ["small3", "full3_fwd"], ids=["reduced_width", "full_width"])
def test_config5_encoder_feeds_the_train_step(gpu, width):
    """BASELINE config 5 (full_width: the GAN at cfg/birds_3stages.yml's own widths, i.e. the configuration as BASELINE.json
    names it on one GPU): CNNRNN.extract_feature at batch 24 on (24, 40, 2048) log-mel with n_frames in 640..2048 sorted
    descending, cap_lens = n_frames // 64 (Audio_to_Image/speech_encoder.py:69-97, extract_audio_feature.py:25-57), feeding
    the StackGAN step.  The embedding is compared with the encoder oracle, the step's losses and images with the
    step oracle fed the ORACLE's embedding."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import CASES, build_nets, make_batch, oracle_dims
    from oracle import speech_encoder_oracle as eorc
    from oracle import stackgan_oracle as orc
    from speech_to_image_translation_without_text_amd import trainer as T
    B = 24
    enc = build_encoder()
    g = torch.Generator().manual_seed(21)
    mel = torch.randn(B, 40, 2048, generator=g) * 20 - 40
    n_frames = torch.sort(torch.randint(640, 2049, (B,), generator=g), descending=True)[0]
    cap_lens = n_frames // 64
    with torch.no_grad():
        _, sent_o = eorc.forward({k: v.clone() for k, v in enc.state_dict().items()}, mel, cap_lens, 512, True)
    case = dict(CASES[width], t=1024, B=B)
    netG, netsD = build_nets(case)
    batch = make_batch(case)
    batch['emb'] = sent_o.clone()
    ostate = orc.TrainState(netG.state_dict(), [d.state_dict() for d in netsD])
    oout = orc.train_step(ostate, batch, oracle_dims(case))
    enc.to(gpu)
    netG.to(gpu)
    for d in netsD:
        d.to(gpu)
    emb = enc.extract_feature(mel.to(gpu), cap_lens)
    assert emb.shape == (B, 1024)
    assert_close(emb, sent_o, rtol=1e-3, atol=1e-5, what="sent_emb B=24")
    tr = T.condGANTrainer(None, None, 256, False)
    tr.build(netG, netsD)
    errD, errG, kl = tr.train_step([t.to(gpu) for t in batch['real']], [t.to(gpu) for t in batch['wrong']],
                                   emb.detach().requires_grad_(True), batch['labels'], batch['noise'].to(gpu),
                                   batch['eps'].to(gpu))
    torch.cuda.synchronize()
    for i in range(3):
        assert_close(tr.fake_imgs[i], oout['fake'][i], rtol=1e-3, atol=1e-4, what="img%d" % i)
    assert_close(float(errD), oout['errD_total'], rtol=1e-3, atol=1e-4, what="errD_total")
    assert_close(float(errG), oout['errG_total'], rtol=1e-3, atol=1e-4, what="errG_total")
    assert_close(float(kl), oout['kl'], rtol=1e-3, atol=1e-5, what="kl")
