"""StackGAN-v2 3-stage G+D train-step benchmark on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full iteration of the reference loop body (trainer.py:536-572, Inception excluded):
G forward, 3x {3 D forwards, 6 BCE terms, backward, all-reduce, Adam}, G update through the updated
Ds {BCE + class-aware + 2*KL, backward, all-reduce, Adam}, EMA; cfg/birds_3stages.yml, batch 24 per
GPU, fp32, synthetic inputs already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMAGE = 1.4264e11       # algorithmic FLOPs per image per train step, branch_num=3
BYTES_PER_STEP_B24 = 28.46e9     # algorithmic HBM bytes per B=24 step
PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 vector = f32-MFMA peak
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-encoder", action="store_true",
                    help="BASELINE config 5: run the speech-encoder front-end (CNN + BiLSTM, eval mode) on synthetic "
                         "log-mel input inside every step and feed its output as the embedding")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the dominant-kernel launches (the command profiled for profiles/*roofline*)")
    ap.add_argument("--roofline-kernel", default=None, choices=[None, "conv", "wgrad"],
                    help="with --roofline-only --math bf16: the convolution kernel (default) or the weight-gradient kernel")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI); gloo only to rehearse the rank logic")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank on cuda:0 (gloo only)")
    ap.add_argument("--cpu-baseline-batch", type=int, default=24)
    ap.add_argument("--graph", type=int, default=0,
                    help="single-GPU step recorded by stream capture after the first warm-up steps: 1 = re-issued as plain "
                         "launches from C, one call per stream piece (s2i_plan_replay); 2 = hipGraphLaunch of each piece "
                         "(measured slower than the Python step on ROCm 7.2, DESIGN.md); 0: enqueue every launch from Python")
    ap.add_argument("--no-side-leg", action="store_true", help="skip the extra bf16x3 measurement of f32 runs (profiling)")
    ap.add_argument("--math", default="f32", choices=["f32", "bf16x3", "bf16x2", "bf16", "bf16p"],
                    help="matrix products of the conv GEMMs: native fp32 MFMA (default), or fp32 operands split into 3 / 2 "
                         "bf16 planes on the bf16 MFMA (DESIGN.md section 9); f32 runs also report bf16x3 beside the value")
    return ap.parse_args()


def synthetic_batch(B, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    batch = dict(emb=torch.randn(B, 1024, device=dev, generator=g),
                 labels=(torch.arange(B, device=dev) % 3).to(torch.int32), real=[], wrong=[])
    for i in range(3):
        s = 64 << i
        batch["real"].append(torch.rand(B, 3, s, s, device=dev, generator=g) * 2 - 1)
        batch["wrong"].append(torch.rand(B, 3, s, s, device=dev, generator=g) * 2 - 1)
    return batch, g


PROFILED_JSON = os.path.join(ROOT, "profiles", "roofline_profiled.json")


def sources_sha16():
    """Hash of the kernel sources: the committed profile numbers name the sources they were collected with, and the line says
    whether they still describe the code that ran."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "speech_to_image_translation_without_text_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def profiled_numbers(key):
    """Numbers of the dominant kernel that cannot be measured live inside this process: the rocprofv3 kernel-trace
    average duration and the PMC HBM traffic of the SAME launch (`bench.py --roofline-only [--math ...]` under
    rocprofv3; tools/profile_roofline.sh).  They live in profiles/roofline_profiled.json next to the CSVs they were
    read from; None when no committed profile covers this configuration."""
    try:
        with open(PROFILED_JSON) as fh:
            ent = json.load(fh).get(key)
    except (OSError, ValueError):
        return None
    if ent is not None:
        ent = dict(ent)
        ent["matches_sources"] = ent.get("sources_sha16") == sources_sha16()   # False: collected with other kernel sources
    return ent


def _time_launch(fn, reps=20):
    for _ in range(3):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def bf16_kernel_roofline(dev, B):
    """bf16 mode (BASELINE config 4): live HIP-event timing of the dominant kernel, conv_bf16_v2_kernel, on its heaviest
    launch of the step: D_NET256's Conv2d(64,128,k4,s2,p1) on the stacked (real | wrong | fake) batch, (3B,128,128,64)
    bf16 NHWC.  Algorithmic bytes per launch (SURVEY.md section 8d rule at 2 B / element): the input read once, the raw
    output written once, the bf16 weights read once."""
    from speech_to_image_translation_without_text_amd import ops
    from speech_to_image_translation_without_text_amd._lib import CONV_K4S2, PACK_PLAIN
    n = 3 * B
    x = torch.randn(n, 128, 128, 64, device=dev).to(torch.bfloat16)
    w = torch.randn(128, 64, 4, 4, device=dev) * 0.03
    packed = ops.pack_weight(w, PACK_PLAIN)
    fn = lambda: ops.conv_any(CONV_K4S2, x, packed, 128, stats=True, groups=3, out_dtype=torch.bfloat16)
    ms = _time_launch(fn)
    flops = 2.0 * (n * 64 * 64) * 128 * (16 * 64)
    abytes = x.numel() * 2 + n * 64 * 64 * 128 * 2 + 16 * 64 * 128 * 2
    gbs = abytes / (ms * 1e-3) / 1e9
    prof = profiled_numbers("bf16_b%d" % B) or {}
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "traffic": prof.get("traffic_bytes"), "traffic_source": prof.get("traffic_source"),
            "frac_profiled": round(abytes / (prof["avg_ns"] * 1e-9) / 1e9 / PEAK_HBM_GBS, 4) if prof.get("avg_ns") else None,
            "profiled_source": prof.get("avg_source"), "profile_matches_sources": prof.get("matches_sources"),
            "kernel": "conv_bf16_v2_kernel<K4S2, CK 32, 4 taps per stage, 66-pixel padded patch rows> (v_mfma_f32_32x32x16_bf16; "
                      "256-pixel x 128-channel tiles, LDS-staged 18x66-pixel input patch, one persistent block per CU "
                      "prefetching its next tile)",
            "algorithmic_bytes": abytes,
            "mfma_tflops": round(flops / (ms * 1e-3) / 1e12, 1), "mfma_frac_of_2500": round(flops / (ms * 1e-3) / 2.5e15, 4),
            "launch": "Conv2d(64,128,k4,s2,p1) on (%d,128,128,64) bf16 NHWC, %.1f MB algorithmic, %.2f GFLOP, %.3f ms"
                      % (n, abytes / 1e6, flops / 1e9, ms),
            "note": "this layer sits at the bf16 ridge (341 FLOP/B against 312): both fractions are reported"}


def bf16_wgrad_roofline(dev, B):
    """bf16 mode: the kernel family with the largest per-step time (profiles/r03_bf16_b48_step_kernel_stats.md):
    igemm_wgrad_b16_kernel (256 x 128 tiles on this launch), on its heaviest launch -- the weight gradient of D_NET256's Conv2d(64,128,k4,s2,p1) over
    the stacked batch: a = (3B,128,128,64) bf16 gathered per tap, g = (3B,64,64,128) bf16.  Algorithmic bytes: both operands
    read once (the fp32 slabs and the OIHW result are < 1 % of that)."""
    from speech_to_image_translation_without_text_amd import ops
    from speech_to_image_translation_without_text_amd._lib import CONV_K4S2
    n = 3 * B
    a = torch.randn(n, 128, 128, 64, device=dev).to(torch.bfloat16)
    g = torch.randn(n, 64, 64, 128, device=dev).to(torch.bfloat16)
    out = torch.zeros(128, 64, 4, 4, device=dev)
    fn = lambda: ops.wgrad_any(CONV_K4S2, a, g, (128, 64, 4, 4), out=out)
    ms = _time_launch(fn)
    flops = 2.0 * (n * 64 * 64) * 128 * (16 * 64)
    abytes = a.numel() * 2 + g.numel() * 2
    gbs = abytes / (ms * 1e-3) / 1e9
    prof = profiled_numbers("bf16_wgrad_b%d" % B) or {}
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "traffic": prof.get("traffic_bytes"), "traffic_source": prof.get("traffic_source"),
            "frac_profiled": round(abytes / (prof["avg_ns"] * 1e-9) / 1e9 / PEAK_HBM_GBS, 4) if prof.get("avg_ns") else None,
            "profiled_source": prof.get("avg_source"), "profile_matches_sources": prof.get("matches_sources"),
            "kernel": "igemm_wgrad_b16_kernel<256,128,4,2> (512-thread blocks; 256 x 256 tiles on the layers with N >= 256) + slab sum "
                      "+ OIHW finish (v_mfma_f32_32x32x16_bf16; 64-pixel stages, pixel-major LDS images read with "
                      "ds_read_b64_tr_b16, fp32 slabs per pixel range)",
            "algorithmic_bytes": abytes,
            "mfma_tflops": round(flops / (ms * 1e-3) / 1e12, 1), "mfma_frac_of_2500": round(flops / (ms * 1e-3) / 2.5e15, 4),
            "launch": "weight gradient of Conv2d(64,128,k4,s2,p1): a (%d,128,128,64) x g (%d,64,64,128) bf16, %.1f MB algorithmic, "
                      "%.2f GFLOP, %.3f ms (incl. the slab reduction and finish launches)" % (n, n, abytes / 1e6, flops / 1e9, ms),
            "note": "largest kernel family of the config-4 step by time (2.1 of 22 ms of kernel time); the convolution kernel's "
                    "roofline is reported beside it as roofline_conv"}


def dominant_kernel_roofline(dev, B, math="f32", which=None):
    """Live HIP-event timing of the dominant kernel: the fp32-MFMA implicit-GEMM convolution, on the
    heaviest single layer of the step (D_NET256 img_code_s16[2]: Conv2d(64,128,k4,s2,p1) on 128x128)."""
    from speech_to_image_translation_without_text_amd import ops
    from speech_to_image_translation_without_text_amd._lib import CONV_K4S2, PACK_PLAIN
    if math == "bf16":
        return bf16_wgrad_roofline(dev, B) if which == "wgrad" else bf16_kernel_roofline(dev, B)
    if math == "bf16p":
        math = "bf16"
    x = torch.randn(B, 128, 128, 64, device=dev)
    w = torch.randn(128, 64, 4, 4, device=dev) * 0.03
    packed = ops.pack_weight(w, PACK_PLAIN)
    flops = 2.0 * (B * 64 * 64) * 128 * (16 * 64)
    for _ in range(3):
        ops.conv_raw(CONV_K4S2, x, None, packed, 128, wR=64, ldw=128, stats=True)
    reps = 20
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        ops.conv_raw(CONV_K4S2, x, None, packed, 128, wR=64, ldw=128, stats=True)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    achieved = flops / (ms * 1e-3) / 1e12
    kernel = {"f32": "igemm_fwd_kernel<128,128,2,2> (v_mfma_f32_32x32x2_f32)",
              "bf16x3": "igemm_fwd_split_kernel<128,128,2,2,3> (6 x v_mfma_f32_32x32x16_bf16 per fp32 product)",
              "bf16x2": "igemm_fwd_split_kernel<128,128,2,2,2> (3 x v_mfma_f32_32x32x16_bf16 per fp32 product)",
              "bf16": "igemm_fwd_split_kernel<128,128,2,2,1> (v_mfma_f32_32x32x16_bf16)"}[math]
    prof = profiled_numbers("%s_b%d" % (math, B)) or {}
    out = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
           "frac": round(achieved / PEAK_F32_TFLOPS, 4),
           # HBM bytes per launch from the committed rocprofv3 PMC passes ((2*FETCH_SIZE + WRITE_SIZE)*1024: the
           # gfx950 correction of MI355X_MICROARCH.md); null when no committed profile covers this batch / math mode
           "traffic": prof.get("traffic_bytes"), "traffic_source": prof.get("traffic_source"),
           "frac_profiled": (round(flops / (prof["avg_ns"] * 1e-9) / 1e12 / PEAK_F32_TFLOPS, 4)
                             if prof.get("avg_ns") else None),
           "profiled_source": prof.get("avg_source"), "profile_matches_sources": prof.get("matches_sources"),
           "kernel": kernel,
           "launch": "Conv2d(64,128,k4,s2,p1) on (%d,128,128,64) NHWC, %.2f GFLOP/launch, %.3f ms" % (B, flops / 1e9, ms)}
    if math != "f32":
        # the split modes execute fp32-equivalent products on the bf16 matrix cores: the fp32 peak is NOT their roof
        out["note"] = ("fp32-equivalent TFLOP/s of a split-bf16 mode; the 157.3 TF fp32 roof does not bound it "
                       "(bf16 MFMA dense peak 2500 TF); frac is reported against the fp32 roof only for comparison")
    return out


MATH_NOTE = {
    "f32": "v_mfma_f32_32x32x2_f32 (fp32 operands, fp32 accumulate)",
    "bf16x3": "each fp32 operand split into 3 bf16 planes, 6 cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulate; "
              "error against fp64 <= that of the native fp32 MFMA path (tools/split_bench.py); opt-in (S2I_MATH_PLANES=3)",
    "bf16": "BASELINE config 4: activations and conv weights stored as bf16 in HBM, v_mfma_f32_32x32x16_bf16 with fp32 "
            "accumulate (patch-staged conv_bf16_kernel / igemm_wgrad_b16_kernel), BatchNorm statistics from the fp32 "
            "accumulators, fp32 master weights / gradients / Adam / EMA; fp32 islands: CA_NET, INIT_STAGE_G.fc, NHWC4 images, "
            "logit heads and losses (S2I_ACT_BF16=1)",
    "bf16p": "operands of the conv GEMMs rounded to bf16 (one plane) on v_mfma_f32_32x32x16_bf16, fp32 accumulate; activations, "
             "BatchNorm statistics, master weights and Adam stay fp32 in HBM (round 1's matrix-product half of config 4; "
             "S2I_MATH_PLANES=1)",
    "bf16x2": "each fp32 operand split into 2 bf16 planes, 3 cross products, fp32 accumulate; ~2^-16 relative per product "
              "(TF32-class, does not hold the 1e-3 parity tolerance end to end); opt-in (S2I_MATH_PLANES=2)",
}


def cpu_baseline(B):
    """The CPU oracle (a restatement of the reference's torch-CPU path) timed on this host: one full
    iteration at branch_num=3, full width."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import CASES, build_nets, make_batch, oracle_dims
    from oracle import stackgan_oracle as orc
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)  # the GPU box gives one GPU's share of the host: 16 cores
    torch.set_num_threads(cores)
    case = dict(CASES["full3_fwd"], B=B)
    netG, netsD = build_nets(case)
    batch = make_batch(case)
    state = orc.TrainState(netG.state_dict(), [d.state_dict() for d in netsD])
    iters = 2
    t0 = time.time()
    for _ in range(iters):
        orc.train_step(state, batch, oracle_dims(case))
    dt = time.time() - t0
    return {"value": round(iters * B / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d full train iterations (G fwd, 3 D updates, G update, EMA), branch_num=3, batch %d, "
                      "torch %s CPU fp32, %.1f s" % (iters, B, torch.__version__, dt)}


def bf16_side_leg(dev, args, model, ops, T, cfg, B=48):
    """One more measurement in the same process: the step in bf16 activation mode at batch 48 (BASELINE config 4)."""
    ops.ACT_BF16 = True
    cfg.TRAIN.BATCH_SIZE = B
    torch.manual_seed(0)
    netG = model.G_NET()
    netG.apply(T.weights_init)
    netsD = []
    for cls in (model.D_NET64, model.D_NET128, model.D_NET256):
        d = cls()
        d.apply(T.weights_init)
        netsD.append(d)
    netG.to(dev)
    for d in netsD:
        d.to(dev)
    tr = T.condGANTrainer(None, None, 256, False, local_rank=dev.index or 0, distributed=False)
    tr.build(netG, netsD)
    batch, gen = synthetic_batch(B, dev, 1)
    noise = torch.empty(B, cfg.GAN.Z_DIM, device=dev)
    eps = torch.empty(B, cfg.GAN.EMBEDDING_DIM, device=dev)

    def step():
        noise.normal_(generator=gen)
        eps.normal_(generator=gen)
        return tr.train_step(batch["real"], batch["wrong"], batch["emb"].detach().requires_grad_(True), batch["labels"], noise,
                             eps)
    # this leg replays the step from its recorded launch plan (trainer.enable_graph): the Python step
    # needs ~13 ms of host time for the ~16 ms of GPU work, so a busy host core shows up in an eager measurement (one run of
    # the pool: 20.2 ms against 16.3 - 16.7); the replay needs ~2 ms.  Eager if the recording fails.
    executor = "eager"
    try:
        tr.enable_graph(warmup=2, executor="plan")
        executor = "plan"
    except Exception:  # noqa: BLE001
        pass
    for _ in range(5):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    losses = [float(v.detach()) for v in out]
    ms = el / args.steps * 1e3
    step_bytes = (836.6e6 * B + 4852.7e6) * 0.5 + 3277.9e6 + 254.9e6
    res = {"value": round(B * args.steps / el, 2), "unit": "images/sec", "ms_per_step": round(ms, 3), "batch": B,
           "dtype": "bf16", "executor": executor, "note": MATH_NOTE["bf16"],
           "step_hbm_frac_of_8TBs": round(step_bytes / (ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
           "losses": {"errD_total": losses[0], "errG_total": losses[1], "kl": losses[2]},
           # the family with the largest per-step time first (weight gradient), the convolution kernel beside it
           "roofline": bf16_wgrad_roofline(dev, B), "roofline_conv": bf16_kernel_roofline(dev, B)}
    del tr, netG, netsD
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(args.backend)

    from speech_to_image_translation_without_text_amd import model, ops, trainer as T
    from speech_to_image_translation_without_text_amd.miscc.config import cfg, cfg_from_file
    ops.MATH_PLANES = {"f32": 0, "bf16x3": 3, "bf16x2": 2, "bf16": 0, "bf16p": 1}[args.math]
    ops.ACT_BF16 = args.math == "bf16"
    if args.roofline_only:
        print(json.dumps({"roofline": dominant_kernel_roofline(dev, args.batch, args.math, args.roofline_kernel)}))
        return
    cfg_from_file(os.path.join(ROOT, "speech_to_image_translation_without_text_amd", "cfg", "birds_3stages.yml"))
    cfg.TRAIN.BATCH_SIZE = args.batch
    B = args.batch

    torch.manual_seed(0)  # identical initial weights on every rank (and equal to the reference's at seed 0)
    netG = model.G_NET()
    netG.apply(T.weights_init)
    netsD = []
    for cls in (model.D_NET64, model.D_NET128, model.D_NET256):
        d = cls()
        d.apply(T.weights_init)
        netsD.append(d)
    netG.to(dev)
    for d in netsD:
        d.to(dev)
    tr = T.condGANTrainer(None, None, 256, False, local_rank=local_rank, distributed=distributed)
    tr.build(netG, netsD)
    if args.graph and not distributed:
        tr.enable_graph(warmup=2, executor="plan" if args.graph == 1 else "graph")

    t_start = time.perf_counter()
    batch, gen = synthetic_batch(B, dev, 1 + rank)
    noise = torch.empty(B, cfg.GAN.Z_DIM, device=dev)
    eps = torch.empty(B, cfg.GAN.EMBEDDING_DIM, device=dev)

    encoder = None
    if args.with_encoder:
        from speech_to_image_translation_without_text_amd.speech_encoder import CNNRNN
        encoder = CNNRNN(40, embedding_dim=1024, nhidden=1024, nsent=1024, bidirectional=True, rnn_layers=1).to(dev).eval()
        mel = torch.randn(B, 40, 2048, device=dev, generator=gen) * 20 - 40
        n_frames = torch.sort(torch.randint(640, 2049, (B,), generator=torch.Generator().manual_seed(1 + rank)),
                              descending=True)[0]
        cap_lens = (n_frames // 64).tolist()

    def one_step():
        noise.normal_(generator=gen)
        eps.normal_(generator=gen)
        if encoder is not None:
            batch["emb"] = encoder.extract_feature(mel, cap_lens)
        emb = batch["emb"].detach().requires_grad_(True)
        return tr.train_step(batch["real"], batch["wrong"], emb, batch["labels"], noise, eps)

    def note(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    def timed(warmup, steps):
        out = None
        for i in range(warmup):
            out = one_step()
            torch.cuda.synchronize()
            note("warm-up step %d done" % i)
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = one_step()
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    elapsed, out = timed(args.warmup, args.steps)
    losses = [float(v) for v in out]
    if not all(abs(v) < 1e6 for v in losses):
        raise RuntimeError("non-finite losses after the timed region: %s" % losses)

    # executed work of ONE step, from the descriptors of the launches themselves (outside the timed region)
    graph_state, tr._graph = tr._graph, None   # the logged step is enqueued from Python (a graph replay runs no host code)
    ops.EXEC_LOG = []
    one_step()
    torch.cuda.synchronize()
    exec_log, ops.EXEC_LOG = ops.EXEC_LOG, None
    tr._graph = graph_state
    exec_flops = sum(r[1] for r in exec_log)
    step_peak_tf = 2500.0 if args.math in ("bf16", "bf16p") else PEAK_F32_TFLOPS

    if rank == 0:
        note("timed region done: %.3f s for %d steps" % (elapsed, args.steps))
        ms = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        step_flops = FLOP_PER_IMAGE * B
        # SURVEY.md section 8d byte rule: activations 836.6 MB / image / step at 4 B per element (2 B in bf16 mode), weights
        # 4852.7 MB per step (bf16 copies: half), Adam 3277.9 MB and EMA 254.9 MB (fp32 in both modes)
        esz = 0.5 if args.math == "bf16" else 1.0
        step_bytes = (836.6e6 * B + 4852.7e6) * esz + 3277.9e6 + 254.9e6
        line = {
            "metric": "StackGAN-v2 3-stage G+D train-step images/sec at 256px",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"f32": "f32", "bf16": "bf16", "bf16p": "bf16 products / f32 storage"}.get(
                args.math, "f32 as " + args.math), "data": "synthetic",
            "config": {"workload": "cfg/birds_3stages.yml: branch_num=3 (64/128/256 px), batch %d per GPU, %s, "
                                   "random-init weights (seed 0, weights_init), synthetic 1024-d embeddings + noise"
                                   % (B, {"bf16": "bf16 activations / weights in HBM, bf16 MFMA with fp32 accumulate, fp32 "
                                                  "BatchNorm statistics / master weights / Adam (BASELINE config 4)",
                                          "bf16p": "bf16 matrix products with fp32 accumulate, fp32 activations / master "
                                                   "weights / Adam"}.get(args.math, "fp32")),
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "speech_encoder_in_step": bool(args.with_encoder), "matrix_products": MATH_NOTE[args.math],
                       "hip_graph": bool(tr._graph is not None and tr._graph.get("graphs") is not None)},
            "step_roofline": {
                # fractions against the dense MFMA peak of the arithmetic the mode computes in (fp32 157.3; bf16 2500)
                # what the matrix cores really deliver: multiply-adds of the launched GEMMs (up-blocks at 4 taps per
                # output parity, c_code folded into a class bias, no D weight gradients in the G update)
                "executed_tflops": round(exec_flops / (ms * 1e-3) / 1e12, 2),
                "executed_flops_frac": round(exec_flops / (ms * 1e-3) / (step_peak_tf * 1e12), 4),
                "flops_peak_tflops": step_peak_tf,
                "executed_gflop_per_image": round(exec_flops / B / 1e9, 2),
                "matrix_launches_per_step": len(exec_log),
                # the reference's own FLOP count for the same step: what a literal execution would need
                "algorithmic_equiv_tflops": round(step_flops / (ms * 1e-3) / 1e12, 2),
                "algorithmic_equiv_frac": round(step_flops / (ms * 1e-3) / (step_peak_tf * 1e12), 4),
                "hbm_frac_of_8TBs": round(step_bytes / (ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
                "algorithmic_gbytes_per_step": round(step_bytes / 1e9, 2),
                "note": "executed_* = 2*M*N*K over the launched GEMM descriptors of one step; algorithmic_equiv_* credits "
                        "the reference's 1.4264e11 FLOP/img/step (SURVEY.md §8d) and is NOT matrix-core utilisation; "
                        "bytes 28.46 GB per B=24 step (fp32 rule of §8d)"},
            "losses": {"errD_total": losses[0], "errG_total": losses[1], "kl": losses[2]},
        }
        line["world_size"] = world
        if distributed:
            line["backend"] = args.backend
            try:
                line["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:  # noqa: BLE001
                line["rccl_version"] = "unavailable: %s" % str(e)[:60]
        if world == 1:
            if args.math == "bf16":
                # config 4: the kernel family with the largest per-step time (weight gradient), the convolution beside it
                line["roofline"] = dominant_kernel_roofline(dev, B, args.math, "wgrad")
                line["roofline_conv"] = dominant_kernel_roofline(dev, B, args.math, "conv")
            else:
                line["roofline"] = dominant_kernel_roofline(dev, B, args.math)
            if args.math == "f32" and not args.no_side_leg:
                # reported beside the value, never as the value: the same step with the split-bf16 matrix products
                try:
                    ops.MATH_PLANES = 3
                    if tr._graph is not None:
                        tr.enable_graph(warmup=2, executor=tr._graph.get("executor", "plan"))   # a recording of its own for this mode
                    el3, out3 = timed(4, args.steps)
                    line["bf16x3_split"] = {"value": round(B * args.steps / el3, 2), "unit": "images/sec",
                                            "ms_per_step": round(el3 / args.steps * 1e3, 3), "note": MATH_NOTE["bf16x3"]}
                except Exception as e:  # noqa: BLE001 - the headline number must not depend on this extra leg
                    line["bf16x3_split"] = {"error": str(e)[:200]}
                finally:
                    ops.MATH_PLANES = 0
            if args.math == "f32" and not args.no_side_leg:
                # BASELINE config 4 (bf16 activations / weights, batch 48 per GPU) beside the headline fp32 value: its own
                # networks, trainer and roofline object; never the value
                try:
                    line["bf16_config4"] = bf16_side_leg(dev, args, model, ops, T, cfg)
                except Exception as e:  # noqa: BLE001
                    line["bf16_config4"] = {"error": str(e)[:200]}
                finally:
                    ops.ACT_BF16 = False
                    cfg.TRAIN.BATCH_SIZE = B
            if not args.no_cpu_baseline:
                note("timing the CPU oracle (bounded sample, batch %d)" % args.cpu_baseline_batch)
                line["cpu_baseline"] = cpu_baseline(args.cpu_baseline_batch)
                note("CPU oracle done")
        print(json.dumps(line))
    if distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
