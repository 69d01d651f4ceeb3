"""Does the train step capture into a hipGraph under the current environment switches?  (dev tool; run one variant per
process: a failing capture can take the process down)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import CASES, build_nets, make_batch
from speech_to_image_translation_without_text_amd import ops, trainer as T
dev = torch.device("cuda:0")
case = dict(CASES['small3'], B=8)
netG, netsD = build_nets(case); batch = make_batch(case)
netG.to(dev); [d.to(dev) for d in netsD]
tr = T.condGANTrainer(None, None, 256, False); tr.build(netG, netsD)
if os.environ.get("PROBE_NUM_D"):
    n = int(os.environ["PROBE_NUM_D"]); tr.netsD, tr.flatsD, tr.num_Ds = tr.netsD[:n], tr.flatsD[:n], n
tr.enable_graph(warmup=2)
b = {k: ([t.to(dev) for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else (v.to(dev) if torch.is_tensor(v) else v)) for k, v in batch.items()}
for it in range(5):
    out = tr.train_step(b['real'], b['wrong'], b['emb'].clone().requires_grad_(True), batch['labels'], b['noise'], b['eps'])
    torch.cuda.synchronize()
    print("step", it, [round(float(o), 5) for o in out], "graph" if tr._graph['graph'] is not None else "eager", flush=True)
