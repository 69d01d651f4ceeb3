"""world_size-2 gloo rehearsal of the data-parallel plumbing (no GPU): rank-0 broadcast of the flat
parameter buffers, per-network gradient all-reduce and the 1/world averaging that the fused Adam
applies (trainer.FlatNet / condGANTrainer._reduce_async).  The reference gets the same semantics from
DistributedDataParallel (trainer.py:167, 192): identical replicas, mean of per-replica gradients."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import CASES, build_nets
    from speech_to_image_translation_without_text_amd import trainer as T
    from speech_to_image_translation_without_text_amd.miscc.config import cfg
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = dict(CASES["small3"], seed=100 + rank)  # different initial weights per rank on purpose
        netG, netsD = build_nets(case)
        cfg.CUDA = False
        tr = T.condGANTrainer(None, None, 256, False, local_rank=rank, distributed=True)
        tr.gpus = [rank]
        assert tr.world == world
        tr.build(netG, netsD)  # flat buffers + broadcast from rank 0
        flats = [tr.flatG] + tr.flatsD
        sums = torch.tensor([float(f.p.double().sum()) for f in flats], dtype=torch.float64)
        gathered = [torch.zeros_like(sums) for _ in range(world)]
        torch.distributed.all_gather(gathered, sums)
        assert all(torch.equal(gathered[0], g) for g in gathered), "replicas differ after the initial broadcast"
        assert torch.equal(tr.flatG.avg, tr.flatG.p), "EMA shadow must start from the broadcast weights"
        # every parameter is a view of the flat buffer, gradients too
        for f in flats:
            for p, o, n in zip(f.params, f.offsets, f.sizes):
                assert p.data_ptr() == f.p.data_ptr() + 4 * o and p.grad.data_ptr() == f.g.data_ptr() + 4 * o
        # per-network all-reduce: each rank contributes rank+1 everywhere; Adam would scale by 1/world
        works = []
        for f in flats:
            f.zero_grad()
            for p in f.params:
                p.grad.add_(float(rank + 1))
            works.append(tr._reduce_async(f))
        for w in works:
            w.wait()
        expect = sum(r + 1 for r in range(world))
        for f in flats:
            for p in f.params:
                assert torch.all(p.grad == expect)
            assert float((f.g * (1.0 / world)).max()) == pytest.approx(expect / world)
        with open(os.path.join(out_dir, "ok%d" % rank), "w") as fh:
            fh.write("ok")
    finally:
        torch.distributed.destroy_process_group()


def test_two_rank_flat_allreduce(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))
