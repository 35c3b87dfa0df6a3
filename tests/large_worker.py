"""Subprocess body of tests/test_gpu_large.py: one forward + backward of a full-size synthetic mini-batch with the
kernel family selected by the environment (the MG_* switches are read once per process).
usage: python tests/large_worker.py <config> <out.npz>"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from molgym_amd.agents.covariant import CovariantAC  # noqa: E402
from molgym_amd.spaces import ActionSpace, ObservationSpace  # noqa: E402
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch  # noqa: E402


def main():
    name, out = sys.argv[1], sys.argv[2]
    cfg = CONFIGS[name]
    torch.manual_seed(0)
    ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']),
                     bag_scale=cfg['bag_scale'], beta=cfg['beta'], device=torch.device('cuda:0'), **MODEL_DEFAULTS)
    data = make_batch(cfg['batch'], cfg['canvas_size'], cfg['zs'], seed=3)
    batch = ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret'])
    ac.theta.grad = torch.zeros_like(ac.theta)
    pred = ac.forward_batch(batch).clone()
    stats = ac.ppo_minibatch(batch, 0.2, 0.5, 0.01)
    torch.cuda.synchronize()
    np.savez(out, pred=pred.cpu().numpy(), stats=stats.cpu().numpy(), grad=ac.theta.grad.cpu().numpy())


if __name__ == '__main__':
    main()
