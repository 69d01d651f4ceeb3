"""Generate tests/golden/encoder.npz from the reference's own Audio_to_Image/speech_encoder.py (CPU, build
container only).  speech_encoder.py needs nothing but torch.  Inputs follow SURVEY.md §8d config 5: log-mel
(B,40,2048) ~ N(0,1)*20-40, n_frames sorted descending, cap_lens = n_frames // 64."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def make_inputs(B=3, seed=11):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 40, 2048, generator=g) * 20 - 40
    n_frames = torch.tensor(sorted([2048, 1400, 640][:B], reverse=True))
    return x, n_frames // 64


def main():
    # the reference is touched only here, when the fixtures are (re)generated in the build container
    sys.path.insert(0, '/root/reference/Audio_to_Image')
    sys.path.insert(0, ROOT)
    import speech_encoder as ref
    from oracle import speech_encoder_oracle as orc
    from make_golden import checksum  # noqa: E402
    torch.manual_seed(0)
    net = ref.CNNRNN(40, embedding_dim=1024, nhidden=1024, nsent=1024, bidirectional=True, rnn_layers=1)
    # non-trivial running statistics, as a trained encoder would have
    g = torch.Generator().manual_seed(5)
    for k, v in net.state_dict().items():
        if k.endswith('running_mean'):
            v.copy_(0.2 * torch.randn(v.shape, generator=g))
        elif k.endswith('running_var'):
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
    net.eval()
    x, lens = make_inputs()
    with torch.no_grad():
        words, sent = net(x, lens)
        ow, osent = orc.forward({k: v.clone() for k, v in net.state_dict().items()}, x, lens, 512, True)
    assert torch.allclose(words, ow, rtol=1e-4, atol=1e-5), float((words - ow).abs().max())
    assert torch.allclose(sent, osent, rtol=1e-4, atol=1e-6)
    np.savez_compressed(os.path.join(HERE, 'encoder.npz'), keys=np.asarray(list(net.state_dict().keys())),
                        checksum=checksum(net.state_dict()), lens=lens.numpy(), words=words.numpy(), sent=sent.numpy())
    print('wrote encoder.npz', words.shape, sent.shape)


if __name__ == '__main__':
    sys.path.insert(0, HERE)
    main()
