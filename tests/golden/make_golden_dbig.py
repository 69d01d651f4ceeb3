"""Golden vectors for D_NET512 / D_NET1024 (reference StackGAN_v2/model.py:555-672), generated from the reference itself.

Run in the build container only (it imports /root/reference):  python tests/golden/make_golden_dbig.py
Writes tests/golden/dbig.npz: for each class, at df=4 / ef=8 and batch 2 -- the seeded input, state_dict checksum,
conditional / unconditional probabilities, x_immediate samples, and the gradient of sum(cond) + sum(uncond) w.r.t. the
image (checksums).  The reference marks D_NET1024 "not test yet"; this pins the restatement to what its code computes.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

CASE = dict(branch=3, gf=4, df=4, ef=8, z=4, t=16, B=2, seed=0, data_seed=1, step=False)


def main():
    cfg, rmodel, rtrainer = mg.import_reference()
    mg.set_cfg(cfg, CASE)
    out = {}
    for size, cls in ((512, rmodel.D_NET512), (1024, rmodel.D_NET1024)):
        torch.manual_seed(CASE['seed'] + size)
        net = cls()
        net.apply(rtrainer.weights_init)
        g = torch.Generator().manual_seed(CASE['data_seed'] + size)
        x = (torch.rand(2, 3, size, size, generator=g) * 2 - 1).requires_grad_(True)
        c = torch.randn(2, CASE['ef'], generator=g)
        out['d%d_state' % size] = mg.checksum(net.state_dict())  # before the forward touches the running statistics
        (cond, uncond), feat = net(x, c)
        (cond.sum() + uncond.sum()).backward()
        out['d%d_cond' % size] = cond.detach().numpy()
        out['d%d_uncond' % size] = uncond.detach().numpy()
        out['d%d_feat' % size] = feat.detach().numpy()
        out['d%d_dx_sum' % size] = np.array([float(x.grad.double().sum()), float(x.grad.double().abs().sum())])
        out['d%d_dx_sample' % size] = x.grad[:, :, ::61, ::53].numpy()
    np.savez_compressed(os.path.join(HERE, 'dbig.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
