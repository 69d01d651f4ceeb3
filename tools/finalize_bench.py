"""Back-to-back timing of the BatchNorm finalize kernels (forward statistics, backward sums) for the partial counts the step
produces (dev tool, GPU box only): 8 - 12 us each including the launch boundary, 23 us for 4608 partial rows."""
import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
from speech_to_image_translation_without_text_amd import ops
from speech_to_image_translation_without_text_amd._lib import ptr, stream, check
lib = ops._lib_ready(); dev = torch.device("cuda:0")
def timeit(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for nparts, G, C in ((2304, 3, 128), (4608, 3, 128), (1152, 3, 256), (768, 1, 32), (3072, 1, 32), (192, 1, 64), (72, 3, 1024), (512, 1, 2048)):
    part = torch.rand(2, nparts, C, device=dev)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev); nbt = torch.zeros((), dtype=torch.long, device=dev)
    coef = torch.empty(G, 4, C, device=dev)
    t0 = timeit(lambda: check(lib.s2i_bn_finalize(ptr(part), nparts, G, C, 100000, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt), 0.1, 1e-5, ptr(coef), stream()), "fin"))
    red2 = torch.empty(G, 2, C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    t1 = timeit(lambda: check(lib.s2i_bn_bwd_finalize(ptr(part), nparts, G, C, 100000, ptr(dg), ptr(db), 1, ptr(red2), stream()), "fin1"))
    print("nparts %5d groups %d C %4d: finalize fwd %6.1f us   bwd %6.1f us  (back to back on one stream)" % (nparts, G, C, t0, t1))
