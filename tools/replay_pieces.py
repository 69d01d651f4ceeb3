"""Where the run-ahead penalty of the launch-plan replay sits (dev tool, GPU box only): HIP events at the piece boundaries of
every step, once with the host synchronised after each step and once with the host free to run ahead.
usage: replay_pieces.py [batch] [bf16]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speech_to_image_translation_without_text_amd import model, ops, trainer as T, _lib
from speech_to_image_translation_without_text_amd.miscc.config import cfg, cfg_from_file
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ops.ACT_BF16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
cfg_from_file(os.path.join(ROOT, "speech_to_image_translation_without_text_amd", "cfg", "birds_3stages.yml"))
torch.manual_seed(0)
netG = model.G_NET(); netG.apply(T.weights_init)
netsD = [c() for c in (model.D_NET64, model.D_NET128, model.D_NET256)]
[d.apply(T.weights_init) for d in netsD]
netG.to(dev); [d.to(dev) for d in netsD]
tr = T.condGANTrainer(None, None, 256, False); tr.build(netG, netsD)
tr.enable_graph(warmup=2)
g = torch.Generator(device=dev).manual_seed(1)
real = [torch.rand(B, 3, 64 << i, 64 << i, device=dev, generator=g) * 2 - 1 for i in range(3)]
wrong = [torch.rand(B, 3, 64 << i, 64 << i, device=dev, generator=g) * 2 - 1 for i in range(3)]
emb = torch.randn(B, 1024, device=dev, generator=g); labels = (torch.arange(B, device=dev) % 3).to(torch.int32)
noise = torch.randn(B, 100, device=dev, generator=g); eps = torch.randn(B, 128, device=dev, generator=g)
for _ in range(5):
    tr.train_step(real, wrong, emb, labels, noise, eps)
torch.cuda.synchronize()
st = tr._graph
lib, pl = _lib.load(), st['plans']
main = torch.cuda.current_stream()
sides = tr._side_streams


def ev():
    return torch.cuda.Event(enable_timing=True)


def one_step(rec):
    e = {k: ev() for k in ("f0", "f1", "g0", "g1")}
    e.update({"d%d_%d" % (i, j): ev() for i in range(3) for j in range(2)})
    e["f0"].record(main)
    lib.s2i_plan_replay(pl['fwd'][0], main.cuda_stream)
    e["f1"].record(main)
    for i in (2, 1, 0):
        sides[i].wait_stream(main)
        e["d%d_0" % i].record(sides[i])
        lib.s2i_plan_replay(pl['d'][i][0], sides[i].cuda_stream)
        e["d%d_1" % i].record(sides[i])
    for i in range(3):
        main.wait_stream(sides[i])
    e["g0"].record(main)
    lib.s2i_plan_replay(pl['g'][0], main.cuda_stream)
    e["g1"].record(main)
    rec.append(e)


for regime in ("sync each step", "run ahead"):
    rec = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(12):
        one_step(rec)
        if regime == "sync each step":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / 12
    acc = {}
    for e in rec[4:]:
        for name, a, b in (("G forward", "f0", "f1"), ("D256 piece", "d2_0", "d2_1"), ("D128 piece", "d1_0", "d1_1"),
                           ("D64 piece", "d0_0", "d0_1"), ("G piece", "g0", "g1"), ("fwd end -> G start", "f1", "g0"),
                           ("step", "f0", "g1")):
            acc.setdefault(name, []).append(e[a].elapsed_time(e[b]))
    print("%s: %.2f ms/step wall | " % (regime, wall) + "  ".join("%s %.2f" % (k, sum(v) / len(v)) for k, v in acc.items()), flush=True)

# every piece replayed ALONE (nothing else on the chip): how long its chain is without contention.  If the discriminator
# phase of the step (fwd end -> G start) is close to the longest chain alone, the phase is bound by that chain's latency; if it
# is close to the SUM of the three, by the chip's throughput.
alone = {}
for name, plan, stream in (("G forward", pl['fwd'][0], main), ("D64 piece", pl['d'][0][0], sides[0]),
                           ("D128 piece", pl['d'][1][0], sides[1]), ("D256 piece", pl['d'][2][0], sides[2]),
                           ("G piece", pl['g'][0], main)):
    ts = []
    for k in range(6):
        torch.cuda.synchronize()
        a, b = ev(), ev()
        a.record(stream)
        lib.s2i_plan_replay(plan, stream.cuda_stream)
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    alone[name] = sum(ts[2:]) / len(ts[2:])
print("alone: " + "  ".join("%s %.2f" % kv for kv in alone.items()) +
      "  | sum of the D pieces %.2f, longest %.2f" % (alone["D64 piece"] + alone["D128 piece"] + alone["D256 piece"],
                                                       max(alone["D64 piece"], alone["D128 piece"], alone["D256 piece"])), flush=True)
