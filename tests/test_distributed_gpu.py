"""Two data-parallel ranks on ONE GPU (gloo collectives, both ranks on cuda:0): the full train step of SURVEY.md §8e --
rank-0 broadcast, per-network gradient all-reduce issued from the discriminator streams, 1/world folded into the fused
Adam -- must keep the replicas bit-identical while they see different data.  (RCCL itself cannot be exercised on a
one-GPU box; the control flow is the same.)"""
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
    from helpers import CASES, build_nets, make_batch
    from speech_to_image_translation_without_text_amd import trainer as T
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        case = dict(CASES["small3"], seed=100 + rank, data_seed=7 + rank)  # different weights AND data per rank
        netG, netsD = build_nets(case)
        netG.to(dev)
        for d in netsD:
            d.to(dev)
        tr = T.condGANTrainer(None, None, 256, False, local_rank=0, distributed=True)
        tr.d_overlap_min = 0         # the reduced-width discriminators are small: split their all-reduce anyway
        tr.build(netG, netsD)
        # D_NET128 / D_NET256 reduce in two chunks (tower + heads from a backward hook, then img_code_s16); at this batch
        # (not a multiple of 8) the three passes are separate calls, so the hook must wait for its third firing
        assert tr._d_split[1] is not None and tr._d_split[2] is not None
        batch = make_batch(case)
        b = {k: ([t.to(dev) for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else
                 (v.to(dev) if torch.is_tensor(v) else v)) for k, v in batch.items()}
        for it in range(2):
            out = tr.train_step(b['real'], b['wrong'], b['emb'].clone().requires_grad_(True), batch['labels'], b['noise'],
                                b['eps'])
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(o).all()) for o in out)
        flats = [tr.flatG] + tr.flatsD
        mine = torch.cat([f.p.detach().cpu().view(-1)[::97] for f in flats] + [tr.flatG.avg.detach().cpu().view(-1)[::97]])
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(gathered, mine)
        assert all(torch.equal(gathered[0], g) for g in gathered), "replicas diverged after two data-parallel steps"
        # the ranks really saw different data: their losses differ
        losses = torch.tensor([float(o) for o in out], dtype=torch.float64)
        gl = [torch.zeros_like(losses) for _ in range(world)]
        torch.distributed.all_gather(gl, losses)
        assert not torch.equal(gl[0], gl[1])
        with open(os.path.join(out_dir, "ok%d" % rank), "w") as fh:
            fh.write("ok")
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_stay_identical(gpu, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))


def _rccl_worker(rank, world, port, out_dir, bf16=False):
    """World-size-1 `nccl` (= RCCL) process group, created before any other GPU call: the collectives of the data-parallel
    step (rank-0 broadcast of parameters and BatchNorm buffers, per-network flat all-reduce issued from the discriminator
    streams, the two-chunk G all-reduce started from a backward hook) run through RCCL itself, and must leave the result
    bit-identical to the non-distributed step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dev = torch.device("cuda:0")
    torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from helpers import CASES, build_nets, make_batch
        from speech_to_image_translation_without_text_amd import ops, trainer as T
        ops.ACT_BF16 = bool(bf16)      # config 4 names an 8-GPU leg: the bf16 re-pack after Adam sits between the all-reduce wait and the next forward
        torch.cuda.set_device(0)
        case = dict(CASES["small3"], B=8)
        batch = make_batch(case)
        b = {k: ([t.to(dev) for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else
                 (v.to(dev) if torch.is_tensor(v) else v)) for k, v in batch.items()}
        finals = []
        for distributed in (True, False):
            netG, netsD = build_nets(case)
            netG.to(dev)
            for d in netsD:
                d.to(dev)
            tr = T.condGANTrainer(None, None, 256, False, local_rank=0, distributed=distributed)
            tr.d_overlap_min = 0     # the reduced-width discriminators are small: split their all-reduce anyway
            tr.build(netG, netsD)
            if distributed:
                assert tr._g_split is not None and 0 < tr._g_split < tr.flatG.total
                assert tr._d_split[2] is not None and 0 < tr._d_split[2] < tr.flatsD[2].total
            for it in range(2):
                out = tr.train_step(b['real'], b['wrong'], b['emb'].clone().requires_grad_(True), batch['labels'],
                                    b['noise'], b['eps'])
            torch.cuda.synchronize()
            finals.append([f.p.clone() for f in [tr.flatG] + tr.flatsD] + [tr.flatG.avg.clone(), tr.flatG.m.clone()]
                          + [torch.stack([o.detach().reshape(()) for o in out])])
        for a, c in zip(*finals):
            assert torch.equal(a, c), "the RCCL path changed the result of a world-size-1 step"
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        with open(os.path.join(out_dir, "ok%d" % rank), "w") as fh:
            fh.write("ok rccl %s" % ver)
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
def test_world_size_one_rccl_group_matches_single_process(gpu, tmp_path, bf16):
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path), bf16), nprocs=1, join=True)
    assert (tmp_path / "ok0").exists()
    print((tmp_path / "ok0").read_text())
