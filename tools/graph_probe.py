"""Replay-vs-eager probe (dev tool, GPU box only): runs the sequence of tests/test_model_gpu.py::
test_graph_replays_interleaved_with_ragged_eager_steps eagerly and with graph replay and prints, per step, whether the
losses agree bit for bit, then the largest difference per state tensor."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import CASES, build_nets
from speech_to_image_translation_without_text_amd import ops, trainer as T
gpu = torch.device("cuda:0")
ops.ACT_BF16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
case = dict(CASES['small3'], B=8)
sizes = [8, 8, 8, 8, 5, 8, 8, 5, 8]
runs = []
for graphed in (False, True):
    netG, netsD = build_nets(case)
    netG.to(gpu); [d.to(gpu) for d in netsD]
    tr = T.condGANTrainer(None, None, 256, False); tr.build(netG, netsD)
    if graphed:
        tr.enable_graph(warmup=2)
    gen = torch.Generator(device=gpu).manual_seed(9)
    losses, states = [], []
    for B in sizes:
        noise = torch.randn(B, case['z'], device=gpu, generator=gen); eps = torch.randn(B, case['ef'], device=gpu, generator=gen)
        real = [torch.rand(B, 3, 64 << i, 64 << i, device=gpu, generator=gen) * 2 - 1 for i in range(3)]
        wrong = [torch.rand(B, 3, 64 << i, 64 << i, device=gpu, generator=gen) * 2 - 1 for i in range(3)]
        emb = torch.randn(B, case['t'], device=gpu, generator=gen)
        out = tr.train_step(real, wrong, emb, [k % 3 for k in range(B)], noise, eps)
        losses.append(torch.stack([o.detach().reshape(()) for o in out]).clone())
        states.append([tr.flatG.p.clone()] + [f.p.clone() for f in tr.flatsD])
    torch.cuda.synchronize()
    runs.append((losses, states))
for k, B in enumerate(sizes):
    la, lb = runs[0][0][k], runs[1][0][k]
    print("step %d B=%d losses equal %s  %s | params max diff %s" % (
        k, B, bool(torch.equal(la, lb)), (la - lb).abs().tolist(),
        [float((a - b).abs().max()) for a, b in zip(runs[0][1][k], runs[1][1][k])]), flush=True)
