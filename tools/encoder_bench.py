"""Speech-encoder front-end alone (BASELINE config 5 shapes); run under rocprofv3 --kernel-trace --stats (dev tool)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd.speech_encoder import CNNRNN  # noqa: E402

dev = torch.device("cuda:0")
B = 24
torch.manual_seed(0)
enc = CNNRNN(40, embedding_dim=1024, nhidden=1024, nsent=1024, bidirectional=True, rnn_layers=1).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
mel = torch.randn(B, 40, 2048, device=dev, generator=g) * 20 - 40
n_frames = torch.sort(torch.randint(640, 2049, (B,), generator=torch.Generator().manual_seed(1)), descending=True)[0]
cap_lens = (n_frames // 64).tolist()
for _ in range(3):
    enc.extract_feature(mel, cap_lens)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    enc.extract_feature(mel, cap_lens)
e1.record()
torch.cuda.synchronize()
print("encoder forward: %.3f ms" % (e0.elapsed_time(e1) / 10))
