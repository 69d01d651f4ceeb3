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
