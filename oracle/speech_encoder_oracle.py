"""CPU oracle for the speech-encoder front-end — TEST INFRASTRUCTURE ONLY (same rules as stackgan_oracle.py).

Restates Audio_to_Image/speech_encoder.py:15-97 (CNNRNN.forward in eval mode) over a {state_dict key: tensor}
mapping with stock torch fp32 ops; the LSTM over packed sequences (speech_encoder.py:84-90) is written out as
the explicit recurrence torch.nn.LSTM computes.  Pinned by tests/golden/encoder.npz (reference outputs).
"""
import torch
import torch.nn.functional as F

# (index in Conv, kernel, stride, padding) of the conv_layer_2d blocks; "P" = MaxPool2d((1,3),(1,2),(0,1))
_LAYERS = [(1, (40, 1), (1, 1), (0, 0)), (2, (1, 3), (1, 1), (0, 1)), "P", (4, (1, 17), (1, 2), (0, 8)),
           (5, (1, 13), (1, 2), (0, 6)), (6, (1, 3), (1, 1), (0, 1)), (7, (1, 9), (1, 2), (0, 4)), "P",
           (9, (1, 3), (1, 1), (0, 1)), (10, (1, 5), (1, 2), (0, 2))]


def _bn_eval(p, prefix, x):
    return F.batch_norm(x, p[prefix + '.running_mean'], p[prefix + '.running_var'], p[prefix + '.weight'],
                        p[prefix + '.bias'], False, 0.1, 1e-5)


def conv_stack(p, x):
    """speech_encoder.py:26-37: (B,1,40,T) -> (B,1024,1,T/64)."""
    x = _bn_eval(p, 'Conv.0', x)
    for layer in _LAYERS:
        if layer == "P":
            x = F.max_pool2d(x, kernel_size=(1, 3), stride=(1, 2), padding=(0, 1))
        else:
            i, k, s, pd = layer
            x = F.conv2d(x, p['Conv.%d.0.weight' % i], stride=s, padding=pd)
            x = F.relu(_bn_eval(p, 'Conv.%d.1' % i, x))
    return x


def lstm_packed(p, x, lens, hidden, bidirectional):
    """nn.LSTM(batch_first) on pack_padded_sequence(x, lens) then pad_packed_sequence(total_length=T):
    per sequence, the forward direction runs t = 0..len-1, the reverse direction t = len-1..0, outputs at
    padded positions are zero (speech_encoder.py:84-88)."""
    B, T, _ = x.shape
    dirs = ["", "_reverse"] if bidirectional else [""]
    out = torch.zeros(B, T, hidden * len(dirs))
    for d, sfx in enumerate(dirs):
        w_ih, w_hh = p['RNN.weight_ih_l0' + sfx], p['RNN.weight_hh_l0' + sfx]
        b = p['RNN.bias_ih_l0' + sfx] + p['RNN.bias_hh_l0' + sfx]
        for bi in range(B):
            h = torch.zeros(hidden)
            c = torch.zeros(hidden)
            steps = range(int(lens[bi]))
            for t in (reversed(steps) if d == 1 else steps):
                g = w_ih @ x[bi, t] + w_hh @ h + b
                i_, f_, g_, o_ = g.split(hidden)
                c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
                h = torch.sigmoid(o_) * torch.tanh(c)
                out[bi, t, d * hidden:(d + 1) * hidden] = h
    return out


def forward(p, x, lens, hidden, bidirectional):
    """CNNRNN.forward (speech_encoder.py:69-97) in eval mode -> (words_emb (B, D*H, T/64), sent_emb (B, D*H))."""
    if x.dim() == 3:
        x = x.unsqueeze(1)
    feat = conv_stack(p, x).squeeze(2).transpose(1, 2)        # (B, T/64, 1024)
    out = lstm_packed(p, feat, lens, hidden, bidirectional)
    return out.transpose(1, 2), out.mean(-2)
