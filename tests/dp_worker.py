"""Subprocess body of tests/test_gpu_dp.py: a short batch_ppo run under torch.distributed (RCCL), one process per GPU.
usage: python -m torch.distributed.run --nproc-per-node N tests/dp_worker.py <out.pt>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molgym_amd import ppo  # noqa: E402
from molgym_amd.env_container import SimpleEnvContainer  # noqa: E402
from molgym_amd.tools import util  # noqa: E402
from molgym_amd.tools.model_util import build_model  # noqa: E402
from molgym_amd.spaces import ActionSpace, ObservationSpace  # noqa: E402
from tests.fake_env import FakeMolEnv  # noqa: E402

ZS = [0, 9, 16]


def main():
    out = sys.argv[1]
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', device_id=dev)
    util.set_seeds(0)  # same model on every rank
    cfg = dict(model='covariant', min_mean_distance=0.8, max_mean_distance=1.8, network_width=64, maxl=4,
               num_cg_levels=3, num_channels_hidden=10, num_channels_per_element=4, num_gaussians=3, bag_scale=5,
               beta=-10)
    ac = build_model(cfg, ObservationSpace(5, ZS), ActionSpace(ZS), dev)
    util.set_seeds(100 + rank)  # every rank drives its own environments with its own random stream
    envs = SimpleEnvContainer([FakeMolEnv(5, ZS, (0, 1 + (4 * rank + i) % 2, 2)) for i in range(4)])
    eval_envs = SimpleEnvContainer([FakeMolEnv(5, ZS, (0, 2, 1))])
    before = ac.theta.detach().clone()
    ppo.batch_ppo(envs=envs, eval_envs=eval_envs, ac=ac, optimizer=util.get_optimizer('adam', 3e-4, ac.parameters()),
                  gamma=1.0, max_num_steps=2 * 16 * world, num_steps_per_iter=16, mini_batch_size=8, clip_ratio=0.2,
                  vf_coef=0.5, entropy_coef=0.01, max_num_train_iters=2, lam=0.97, target_kl=1e9, gradient_clip=0.5,
                  eval_freq=1, num_eval_episodes=1, device=dev)
    theta = ac.theta.detach()
    gathered = [torch.empty_like(theta) for _ in range(world)]
    dist.all_gather(gathered, theta)
    if rank == 0:
        torch.save({'moved': not torch.equal(before, theta), 'finite': bool(torch.isfinite(theta).all()),
                    'replicas_equal': all(torch.equal(g, theta) for g in gathered), 'world': world}, out)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
