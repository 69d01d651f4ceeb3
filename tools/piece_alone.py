"""One stream piece of the recorded step replayed ALONE, for a kernel trace without contention (dev tool, GPU box only).
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/piece_alone.py <batch> <f32|bf16> <fwd|d0|d1|d2|g>
       python3 tools/piece_alone.py summarise DIR/.../t_kernel_trace.csv
The replays are the last kernels of the run: the summary takes the last reps x (kernels of the plan) rows of the trace (the
count is written to gpurun_out/piece_alone.count by the run)."""
import collections
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS = 4
COUNT_FILE = os.path.join(ROOT, "gpurun_out", "piece_alone.count")


def summarise(path):
    nk, what = open(COUNT_FILE).read().split(None, 1)
    nk = int(nk)
    rows = []
    with open(path) as fp:
        for r in csv.DictReader(fp):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    seg = rows[-nk * REPS:]

    def short(n):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        return re.sub(r"\(.*$", "", n)[:64]

    by = collections.OrderedDict()
    for s, e, n in seg:
        d = by.setdefault(short(n), [0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
    tot = sum(v[1] for v in by.values()) / REPS
    wall = sum(seg[(k + 1) * nk - 1][1] - seg[k * nk][0] for k in range(REPS)) / 1e3 / REPS
    gemm = sum(v[1] for k, v in by.items() if re.search(r"igemm|conv_bf16|wgrad_k3|rgb_|small_n|thin_|n4_tile", k)) / REPS
    print("%s alone: %d kernels per replay, kernel time %.0f us, wall %.0f us (idle %.0f us); matrix kernels %.0f us"
          % (what.strip(), nk, tot, wall, wall - tot, gemm))
    for k, (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]:
        print("   %8.1f us %5.1f x %7.1f  %s" % (us / REPS, c / REPS, us / c, k))


if len(sys.argv) > 1 and sys.argv[1] == "summarise":
    summarise(sys.argv[2])
    sys.exit(0)

import torch  # noqa: E402
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import model, ops, trainer as T, _lib  # noqa: E402
from speech_to_image_translation_without_text_amd.miscc.config import cfg_from_file  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ops.ACT_BF16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
which = sys.argv[3] if len(sys.argv) > 3 else "d2"
cfg_from_file(os.path.join(ROOT, "speech_to_image_translation_without_text_amd", "cfg", "birds_3stages.yml"))
torch.manual_seed(0)
netG = model.G_NET()
netG.apply(T.weights_init)
netsD = [c() for c in (model.D_NET64, model.D_NET128, model.D_NET256)]
for d in netsD:
    d.apply(T.weights_init)
netG.to(dev)
for d in netsD:
    d.to(dev)
tr = T.condGANTrainer(None, None, 256, False)
tr.build(netG, netsD)
tr.enable_graph(warmup=2)
g = torch.Generator(device=dev).manual_seed(1)
real = [torch.rand(B, 3, 64 << i, 64 << i, device=dev, generator=g) * 2 - 1 for i in range(3)]
wrong = [torch.rand(B, 3, 64 << i, 64 << i, device=dev, generator=g) * 2 - 1 for i in range(3)]
emb = torch.randn(B, 1024, device=dev, generator=g)
labels = (torch.arange(B, device=dev) % 3).to(torch.int32)
noise = torch.randn(B, 100, device=dev, generator=g)
eps = torch.randn(B, 128, device=dev, generator=g)
for _ in range(5):
    tr.train_step(real, wrong, emb, labels, noise, eps)
torch.cuda.synchronize()
pl = tr._graph['plans']
lib = _lib.load()
plan, counts = {"fwd": pl['fwd'], "g": pl['g'], "d0": pl['d'][0], "d1": pl['d'][1], "d2": pl['d'][2]}[which]
stream = torch.cuda.current_stream() if which in ("fwd", "g") else tr._side_streams[int(which[1])]
os.makedirs(os.path.dirname(COUNT_FILE), exist_ok=True)
with open(COUNT_FILE, "w") as fp:
    fp.write("%d %s piece, batch %d, %s\n" % (counts[0], which, B, "bf16" if ops.ACT_BF16 else "fp32"))
assert counts[1] == 0 and counts[2] == 0, "memset / memcpy nodes do not show in a kernel trace: %r" % (counts,)
for _ in range(REPS):
    torch.cuda.synchronize()
    lib.s2i_plan_replay(plan, stream.cuda_stream)
torch.cuda.synchronize()
print("replayed %s %d times: %d kernels each" % (which, REPS, counts[0]))
