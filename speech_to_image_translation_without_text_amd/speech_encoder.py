"""Speech-encoder front-end on the MI355X kernels: the surface of the reference's
Audio_to_Image/speech_encoder.py:15-97 (`CNNRNN`), inference only (BASELINE config 5: the encoder runs in
.eval() mode in front of the GAN step and feeds the 1024-d embedding; calling convention
Audio_to_Image/extract_audio_feature.py:25-57).

Same constructor signature, same `state_dict` keys (`Conv.*`, `RNN.*`) and the same RNG consumption as the
reference, so seeded weights and checkpoints are interchangeable.  What runs is different:
  * eval-mode BatchNorms are folded into the convolutions (one scale per output channel into the weights, one
    bias), the leading BatchNorm2d(1) too;
  * log-mel input (B, 40, T) is laid out NHWC [B, 1, T, 40]: the (40 x 1) conv is a 1x1 conv over 40 channels,
    the temporal convs are the implicit-GEMM kernel's 1-D kind with fused bias + ReLU;
  * the LSTM's input projections for all steps and both directions are ONE GEMM; each recurrent step is a small
    GEMM + a fused gate kernel that applies the packed-sequence rule (per-sequence length, reverse direction
    starting at len-1, padded outputs zero) without packing anything.
Training the encoder (JEL loss etc.) is out of scope: forward in training mode raises.
"""
import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import ACT_RELU, CONV_1D, CONV_K1, PACK_PLAIN, check, ptr, stream


def conv_layer_2d(in_channel, out_channel, kernel_size, stride, padding):
    return nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size, stride, padding, bias=False),
                         nn.BatchNorm2d(out_channel), nn.ReLU(inplace=True))


class CNNRNN(nn.Module):
    def __init__(self, n_filters, embedding_dim=1024, drop_prob=0.5, nhidden=1024, rnn_layers=1,
                 bidirectional=False, nsent=1024):
        super().__init__()
        self.bidirectional = bidirectional
        self.num_direction = 2 if bidirectional else 1
        self.embedding_dim = embedding_dim
        self.nhidden = nhidden // self.num_direction
        self.rnn_layers = rnn_layers
        self.drop_prob = drop_prob
        self.nsent = nsent // self.num_direction
        self.Conv = nn.Sequential(
            nn.BatchNorm2d(1),
            conv_layer_2d(1, 64, (n_filters, 1), (1, 1), (0, 0)),
            conv_layer_2d(64, 64, (1, 3), (1, 1), (0, 1)),
            nn.MaxPool2d(kernel_size=(1, 3), stride=(1, 2), padding=(0, 1)),
            conv_layer_2d(64, 128, (1, 17), (1, 2), (0, 8)),
            conv_layer_2d(128, 256, (1, 13), (1, 2), (0, 6)),
            conv_layer_2d(256, 256, (1, 3), (1, 1), (0, 1)),
            conv_layer_2d(256, 512, (1, 9), (1, 2), (0, 4)),
            nn.MaxPool2d(kernel_size=(1, 3), stride=(1, 2), padding=(0, 1)),
            conv_layer_2d(512, 512, (1, 3), (1, 1), (0, 1)),
            conv_layer_2d(512, 1024, (1, 5), (1, 2), (0, 2)))
        self.RNN = nn.LSTM(self.embedding_dim, self.nhidden, num_layers=self.rnn_layers, batch_first=True,
                           bidirectional=self.bidirectional, dropout=self.drop_prob)
        self.apply(self.weights_init)
        self._prepared = None

    @staticmethod
    def weights_init(m):
        classname = m.__class__.__name__
        if classname.find('Conv') != -1:
            m.weight.data.normal_(0.0, 0.02)
        elif classname.find('BatchNorm') != -1:
            m.weight.data.normal_(1.0, 0.02)
            m.bias.data.fill_(0)
        elif classname.find('Linear') != -1:
            m.weight.data.normal_(0.0, 0.02)
            if m.bias is not None:
                m.bias.data.fill_(0.0)

    # -- weight preparation: BatchNorm folding + packing, redone when any tensor changes -------------------
    def _signature(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    @torch.no_grad()
    def _prepare(self):
        sig = self._signature()
        if self._prepared is not None and self._prepared[0] == sig:
            return self._prepared[1]
        if self.rnn_layers != 1:
            raise _lib.S2IError("CNNRNN on this path supports rnn_layers=1 (the reference's default)")
        eps = 1e-5
        bn0 = self.Conv[0]
        a0 = (bn0.weight / torch.sqrt(bn0.running_var + eps)).reshape(())
        b0 = (bn0.bias - bn0.running_mean * a0).reshape(())
        layers = []
        first = True
        for m in self.Conv:
            if isinstance(m, nn.MaxPool2d):
                layers.append(("pool",))
            elif isinstance(m, nn.Sequential):
                conv, bn = m[0], m[1]
                s = bn.weight / torch.sqrt(bn.running_var + eps)
                w = conv.weight * s.view(-1, 1, 1, 1)
                bias = bn.bias - bn.running_mean * s
                if first:  # fold the scalar input BatchNorm: conv(a0*x + b0) = a0*conv(x) + b0*sum(w)
                    bias = bias + b0 * w.sum(dim=(1, 2, 3))
                    w = w * a0
                    w2 = w.reshape(w.shape[0], w.shape[2]).contiguous()  # (64, 1, 40, 1) -> (64, 40)
                    layers.append(("k1", ops.pack_weight(w2, PACK_PLAIN), bias.contiguous(), w.shape[0]))
                    first = False
                else:
                    k, st, pd = conv.kernel_size[1], conv.stride[1], conv.padding[1]
                    layers.append(("c1d", ops.pack_weight(w.contiguous(), PACK_PLAIN), bias.contiguous(), w.shape[0],
                                   (k, st, pd)))
        rnn = self.RNN
        sfx = ["", "_reverse"][:self.num_direction]
        w_ih = torch.cat([getattr(rnn, "weight_ih_l0" + s_) for s_ in sfx], 0).contiguous()       # (D*4H, E)
        b_ih = torch.cat([getattr(rnn, "bias_ih_l0" + s_) + getattr(rnn, "bias_hh_l0" + s_) for s_ in sfx], 0)
        w_hh = [ops.pack_weight(getattr(rnn, "weight_hh_l0" + s_).contiguous(), PACK_PLAIN) for s_ in sfx]
        w_hh_raw = [getattr(rnn, "weight_hh_l0" + s_).detach().contiguous() for s_ in sfx]
        prep = dict(layers=layers, w_ih=ops.pack_weight(w_ih, PACK_PLAIN), b_ih=b_ih.contiguous(), w_hh=w_hh,
                    w_hh_raw=w_hh_raw)
        self._prepared = (sig, prep)
        return prep

    def extract_feature(self, x, lens):
        return self.forward(x, lens)[1]

    @torch.no_grad()
    def forward(self, x, cap_lens):
        """x: (B, 40, T) or (B, 1, 40, T) log-mel; cap_lens: (B,) valid LSTM steps, sorted descending as the
        reference requires.  Returns (words_emb (B, D*H, T/64), sent_emb (B, nsent*D))."""
        if self.training:
            raise _lib.S2IError("CNNRNN: only the inference (.eval()) path is built on the MI355X kernels")
        lib = _lib.load()
        _lib.require_device()
        if x.dim() == 4:
            x = x[:, 0]
        B, F_, T = x.shape
        prep = self._prepare()
        h = ops.ToNHWC.apply(x.reshape(B, F_, 1, T).contiguous(), F_)  # [B, 1, T, 40]
        for layer in prep["layers"]:
            if layer[0] == "pool":
                Bh, Hh, Wh, Ch = h.shape
                out = torch.empty((Bh, Hh, Wh // 2, Ch), dtype=torch.float32, device=h.device)
                check(lib.s2i_maxpool_w3s2(ptr(h), Bh, Hh, Wh, Ch, ptr(out), stream()), "s2i_maxpool_w3s2")
                h = out
            elif layer[0] == "k1":
                _, packed, bias, n = layer
                h, _, _ = ops.conv_raw(CONV_K1, h, None, packed, n, wR=packed.shape[1], ldw=packed.shape[2], bias=bias,
                                       act=ACT_RELU)
            else:
                _, packed, bias, n, geom = layer
                h, _, _ = ops.conv_raw(CONV_1D, h, None, packed, n, wR=packed.shape[1], ldw=packed.shape[2], bias=bias,
                                       act=ACT_RELU, conv1d=geom)
        Bh, _, L, E = h.shape                                           # [B, 1, T/64, 1024]
        lens_host = [int(v) for v in (cap_lens.tolist() if torch.is_tensor(cap_lens) else cap_lens)]
        if len(lens_host) != B or max(lens_host) > L or min(lens_host) < 1:
            raise _lib.S2IError("CNNRNN: cap_lens must hold B values in [1, %d]" % L)
        lens_dev = torch.tensor(lens_host, dtype=torch.int32, device=h.device)
        D, Hd = self.num_direction, self.nhidden
        w_ih = prep["w_ih"]
        xproj, _, _ = ops.conv_raw(CONV_K1, h.view(B, 1, L, E), None, w_ih, D * 4 * Hd, wR=w_ih.shape[1], ldw=w_ih.shape[2],
                                   bias=prep["b_ih"])                   # [B, 1, L, D*4H]
        out = torch.zeros((B, L, D * Hd), dtype=torch.float32, device=h.device)
        if B <= 32 and Hd % 8 == 0 and Hd <= 512:
            # one launch per time step for both directions: recurrent projection + cell fused (s2i_lstm_step)
            hbuf = torch.zeros((2, D, B, Hd), dtype=torch.float32, device=h.device)
            cs = torch.zeros((D, B, Hd), dtype=torch.float32, device=h.device)
            raw = prep["w_hh_raw"]
            for step in range(max(lens_host)):
                check(lib.s2i_lstm_step(ptr(xproj), D * 4 * Hd, ptr(raw[0]), ptr(raw[-1]), ptr(lens_dev), B, L, Hd, D, step,
                                        ptr(hbuf[step & 1]), ptr(hbuf[(step + 1) & 1]), ptr(cs), ptr(out), D * Hd, stream()),
                      "s2i_lstm_step")
        else:
            for d in range(D):
                hs = torch.zeros((B, 1, 1, Hd), dtype=torch.float32, device=h.device)
                cs = torch.zeros((B, Hd), dtype=torch.float32, device=h.device)
                w_hh = prep["w_hh"][d]
                for step in range(max(lens_host)):
                    hproj, _, _ = ops.conv_raw(CONV_K1, hs, None, w_hh, 4 * Hd, wR=w_hh.shape[1], ldw=w_hh.shape[2])
                    check(lib.s2i_lstm_cell(ptr(xproj) + 4 * d * 4 * Hd, D * 4 * Hd, ptr(hproj), ptr(lens_dev), B, L, Hd,
                                            step, d, ptr(hs), ptr(cs), ptr(out) + 4 * d * Hd, D * Hd, stream()),
                          "s2i_lstm_cell")
        sent = torch.empty((B, D * Hd), dtype=torch.float32, device=h.device)
        check(lib.s2i_time_mean(ptr(out), B, L, D * Hd, ptr(sent), stream()), "s2i_time_mean")
        return out.transpose(1, 2), sent.view(-1, self.nsent * self.num_direction)
