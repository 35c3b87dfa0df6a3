"""Drop-in boundary: the `molgym` shim resolves the reference's import names (scripts/run.py:6-13) to the HIP hot
path, the agents need nothing of the observation space beyond what the REFERENCE's ObservationSpace offers
(`zs`, `canvas_space.size`; spaces.py:96-107), checkpoints speak the reference's state_dict keys, and a scaled-down
scripts/run.py main loop (batch_ppo with rollouts, PPO updates, evaluation, ModelIO / InfoSaver / RolloutSaver
output) runs end to end on the device."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZS = [0, 9, 16]
CONFIG = dict(model='covariant', min_mean_distance=0.8, max_mean_distance=1.8, network_width=128, maxl=4,
              num_cg_levels=3, num_channels_hidden=10, num_channels_per_element=4, num_gaussians=3, bag_scale=5,
              beta=-10)


class _RefCanvasSpace:
    def __init__(self, size):
        self.size = size


class RefLikeObservationSpace:
    """ONLY the attributes the reference's ObservationSpace has that an agent may touch (spaces.py:96-107); anything
    else raises AttributeError, like it would on the real class."""

    def __init__(self, canvas_size, zs):
        self.zs = zs
        self.canvas_space = _RefCanvasSpace(canvas_size)

    def parse(self, observation):  # returns ase.Atoms in the reference; the HIP agents never call it
        raise AssertionError('agent went through ObservationSpace.parse (ase round trip)')


class RefLikeActionSpace:
    def __init__(self, zs):
        self.zs = zs


def test_shim_import_names_of_run_py():
    from molgym.env_container import SimpleEnvContainer  # noqa: F401
    from molgym.ppo import batch_ppo, train  # noqa: F401
    from molgym.spaces import ActionSpace, ObservationSpace  # noqa: F401
    from molgym.tools import util
    from molgym.tools.model_util import ModelIO, build_model  # noqa: F401
    from molgym.agents.base import AbstractActorCritic
    from molgym.agents.covariant.agent import CovariantAC
    from molgym.agents.internal.agent import SchNetAC
    import molgym_amd.ppo
    assert train is molgym_amd.ppo.train and batch_ppo is molgym_amd.ppo.batch_ppo
    assert issubclass(CovariantAC, AbstractActorCritic) and issubclass(SchNetAC, AbstractActorCritic)
    assert callable(util.set_seeds) and callable(util.get_optimizer) and callable(util.count_vars)


@pytest.mark.skipif(not os.path.isdir('/root/reference/molgym'), reason='reference checkout not present')
def test_shim_layers_over_a_reference_checkout():
    """with MOLGYM_REFERENCE set, everything off the hot path comes from the reference's own files"""
    code = ('import molgym, molgym.ppo, molgym.version, molgym.tools.model_util as mu;'
            'print(molgym.version.__file__); print(molgym.ppo.train.__module__); print(mu.build_model.__module__)')
    env = dict(os.environ, MOLGYM_REFERENCE='/root/reference', PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ['/root/reference/molgym/version.py', 'molgym_amd.ppo', 'molgym_amd.tools.model_util']


def test_agent_builds_on_the_reference_observation_space_surface():
    from molgym.tools.model_util import build_model
    ac = build_model(CONFIG, RefLikeObservationSpace(7, ZS), RefLikeActionSpace(ZS), torch.device('cpu'))
    assert ac.training and hasattr(ac, 'evaluate_actions')
    obs = (((1, (0.0, 0.0, 0.0)), (0, (0.0, 0.0, 0.0)), (2, (1.0, 2.0, 3.0))) + ((0, (0.0, 0.0, 0.0)), ) * 4, (0, 2, 1))
    # focus indexes the non-null items in canvas order: index 1 is the atom at (1, 2, 3)
    element, position = ac.to_action_space(np.array([1, 2, 1.5, 0.0, 0.0, 1.0]), obs)
    assert element == 2 and np.allclose(position, (1.0, 2.0, 4.5))
    assert ac.to_action_space(np.array([0, 1, 1.5, 0.0, 0.0, 1.0]), (((0, (0.0, 0.0, 0.0)), ) * 7, (0, 2, 1)))[1] == (0.0, 0.0, 0.0)
    with pytest.raises(RuntimeError):  # no CPU fallback for the arithmetic
        ac.step([obs], np.array([[0, 1, 1.5, 0.0, 0.0, 1.0]]))
    # re-assignable spaces (run.py:53-54)
    ac.observation_space, ac.action_space = RefLikeObservationSpace(7, ZS), RefLikeActionSpace(ZS)


def test_state_dict_speaks_reference_keys(tmp_path):
    from molgym.tools.model_util import build_model
    from molgym_amd import layout
    torch.manual_seed(1)
    a = build_model(CONFIG, RefLikeObservationSpace(7, ZS), RefLikeActionSpace(ZS), torch.device('cpu'))
    b = build_model(CONFIG, RefLikeObservationSpace(7, ZS), RefLikeActionSpace(ZS), torch.device('cpu'))
    sd = a.state_dict()
    assert list(sd) == list(layout.slots(len(ZS), 128, 3))  # the reference module's tensor names, in order
    assert sd['phi_focus.layers.0.weight'].shape == (128, 144) and sd['distance_log_stds'].shape == (3, )
    assert sd['cg_model.cormorant_cg.atom_levels.2.cat_mix.weights.4'].shape == (12, 310, 2)
    assert not torch.equal(a.theta, b.theta)
    torch.save(sd, tmp_path / 'sd.pt')
    # a reference checkpoint also carries non-trainable cormorant buffers: tolerated; real strangers are not
    extra = dict(torch.load(tmp_path / 'sd.pt'))
    extra['cg_model.cormorant_cg.edge_levels.0.mask_layer.soft_cut_rad'] = torch.zeros(1)
    res = b.load_state_dict(extra)
    assert torch.equal(a.theta, b.theta) and not res.missing_keys and not res.unexpected_keys
    with pytest.raises(RuntimeError):
        b.load_state_dict({**sd, 'phi_new.weight': torch.zeros(1)})
    with pytest.raises(RuntimeError):
        b.load_state_dict({k: v for k, v in sd.items() if k != 'distance_log_stds'})
    bad = dict(sd)
    bad['distance_log_stds'] = torch.zeros(4)
    with pytest.raises(RuntimeError, match='size mismatch'):
        b.load_state_dict(bad)
    assert b.load_state_dict({'theta': a.theta.detach().clone()}).missing_keys == []
    # internal agent: same contract
    c = build_model(dict(CONFIG, model='internal'), RefLikeObservationSpace(7, ZS), RefLikeActionSpace(ZS), torch.device('cpu'))
    assert 'embedding_fn.interactions.2.cfconv.in2f.weight' in c.state_dict() and 'critic.layers.2.bias' in c.state_dict()


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['covariant', 'internal'])
def test_run_py_main_loop_end_to_end(built_lib, tmp_path, model):
    """scripts/run.py:60-121 in small: build_model -> batch_ppo (rollouts, updates, evaluation, saving) -> reload"""
    from molgym.env_container import SimpleEnvContainer
    from molgym.ppo import batch_ppo
    from molgym.tools import util
    from molgym.tools.model_util import ModelIO, build_model
    from tests.fake_env import FakeMolEnv
    for d in ('models', 'data', 'results'):
        os.makedirs(tmp_path / d)
    util.set_seeds(0)
    device = util.init_device('cuda')
    obs_space, act_space = RefLikeObservationSpace(5, ZS), RefLikeActionSpace(ZS)
    ac = build_model(dict(CONFIG, model=model, network_width=64), obs_space, act_space, device)
    n_params = util.count_vars(ac)
    assert n_params == ac.theta.numel() > 1e5
    envs = SimpleEnvContainer([FakeMolEnv(5, ZS, (0, 1 + i % 2, 2)) for i in range(4)])
    eval_envs = SimpleEnvContainer([FakeMolEnv(5, ZS, (0, 2, 1))])
    handler = ModelIO(directory=str(tmp_path / 'models'), tag='t', keep=False)
    before = ac.theta.detach().clone()
    batch_ppo(envs=envs, eval_envs=eval_envs, ac=ac,
              optimizer=util.get_optimizer('adam', 3e-4, ac.parameters()), gamma=1.0, start_num_steps=0,
              max_num_steps=48, num_steps_per_iter=16, mini_batch_size=8, clip_ratio=0.2, vf_coef=0.5,
              entropy_coef=0.01, max_num_train_iters=2, lam=0.97, target_kl=1e9, gradient_clip=0.5, eval_freq=1,
              model_handler=handler, save_freq=1, num_eval_episodes=1,
              rollout_saver=util.RolloutSaver(directory=str(tmp_path / 'data'), tag='t'), save_train_rollout=True,
              save_eval_rollout=True, info_saver=util.InfoSaver(directory=str(tmp_path / 'results'), tag='t'),
              device=device)
    assert not torch.equal(before, ac.theta.detach()) and torch.isfinite(ac.theta).all()
    # outputs the reference's analysis tools read (tools/analysis.py:8-47): JSON lines with these keys, pickled buffers
    opt = [json.loads(l) for l in open(tmp_path / 'results' / 't_opt.txt')]
    assert len(opt) == 3 and {'policy_loss', 'entropy_loss', 'vf_loss', 'total_loss', 'approx_kl', 'clip_fraction',
                              'grad_norm', 'num_opt_steps', 'time', 'total_num_steps'} <= set(opt[0])
    assert all(np.isfinite(r['total_loss']) and r['num_opt_steps'] == 2 for r in opt)
    train = [json.loads(l) for l in open(tmp_path / 'results' / 't_train.txt')]
    assert {'return_mean', 'return_std', 'episode_length_mean', 'value_mean', 'logp_std', 'total_num_steps'} <= set(train[0])
    assert os.path.exists(tmp_path / 'results' / 't_eval.txt')
    buf = pickle.load(open(tmp_path / 'data' / 't_steps-0_train.pkl', 'rb'))
    assert len(buf.obs_buf) == 16 == len(buf.adv_buf) and buf.is_finished()
    # only the latest model is kept; it reloads into an agent that steps
    assert os.listdir(tmp_path / 'models') == ['t_steps-48.model']
    again, steps = handler.load_latest(device=device)
    assert steps == 48 and torch.equal(again.theta, ac.theta)
    again.observation_space, again.action_space = obs_space, act_space
    again.training = False
    out = again.step(envs.reset())
    assert len(out['actions']) == 4 and torch.isfinite(out['logp']).all()
