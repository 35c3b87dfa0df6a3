"""Debug tool (GPU box): wall time of molgym_amd.ppo.batch_rollout per environment step with zero-cost fake environments
(tests/fake_env.py) -- the loop's own overhead (policy step on resident canvases, buffer stores, resets, re-uploads).
usage: python tools/rollout_loop_bench.py [num_envs] [iterations]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from molgym_amd import ppo  # noqa: E402
from molgym_amd.agents.covariant import CovariantAC  # noqa: E402
from molgym_amd.buffer import PPOBufferContainer  # noqa: E402
from molgym_amd.env_container import SimpleEnvContainer  # noqa: E402
from molgym_amd.spaces import ActionSpace, ObservationSpace  # noqa: E402
from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS  # noqa: E402
from tests.fake_env import FakeMolEnv  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 140
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 21
cfg = CONFIGS['cfg2']
torch.manual_seed(0)
ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']), bag_scale=cfg['bag_scale'],
                 beta=cfg['beta'], device=torch.device('cuda'), **MODEL_DEFAULTS)
envs = SimpleEnvContainer([FakeMolEnv(cfg['canvas_size'], cfg['zs'], (0, 1, 6)) for _ in range(E)])
for rep in range(3):
    container = PPOBufferContainer(size=E, gamma=0.99, lam=0.97)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ppo.batch_rollout(ac, envs, container, num_steps=E * iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{E} environments x {iters} steps: {dt * 1e3:.1f} ms -> {dt / iters * 1e3:.3f} ms per step of all environments, '
          f'{E * iters / dt:.0f} environment steps/s')
