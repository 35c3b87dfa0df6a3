"""The reference's own agent tests (tests/agents/covariant/test_agent.py:43-123) on the CPU oracle, same molecules
(tests/golden/test_agent_molecules.json), same configuration, Wigner matrices from sympy: the conditioned orientation
coefficients of a rotated canvas are the Wigner-rotated coefficients of the original one.  This is the only pin the
reference holds on the Cormorant encoder, whose source is not on disk."""
import numpy as np
import pytest
import torch

from oracle.covariant_ref import CovariantACRef
from tests.helpers import (atomic_scalars_of, euler_rotation, load_test_agent_molecules, molecule_observation,
                           wigner_d_sympy)

ANGLES = (0.3, 1.1, -0.7)


@pytest.fixture(scope='module')
def setup_ref():
    mols = load_test_agent_molecules()
    su = mols['setup']
    torch.manual_seed(0)
    ref = CovariantACRef(zs=su['zs'], canvas_size=su['canvas_size'], min_max_distance=tuple(su['min_max_distance']),
                         network_width=su['network_width'], maxl=4, num_cg_levels=3, num_channels_hidden=10,
                         num_channels_per_element=4, num_gaussians=3, bag_scale=su['bag_scale'], beta=su['beta']).double()
    return mols, su, ref


def _normalised(cond):
    k = sum((p.sum(dim=-3)**2).sum(dim=(-1, -2)) for p in cond).clamp(min=1e-10).sqrt().view(-1, 1, 1, 1)
    return [p / k for p in cond]


@pytest.mark.parametrize('name', ['h2o', 'ch3', 'ch4'])
def test_coefficients_rotate_with_the_canvas(setup_ref, name):
    mols, su, ref = setup_ref
    R, D = euler_rotation(*ANGLES), wigner_d_sympy(*ANGLES)
    act = np.array([[1, 1, 1.2, 0.3, 0.4, 0.5]])
    with torch.no_grad():
        a = ref.step([molecule_observation(mols[name], su)], act, dtype=torch.float64, return_internals=True)
        b = ref.step([molecule_observation(mols[name], su, R)], act, dtype=torch.float64, return_internals=True)
    ca, cb = _normalised(a['cond_cov']), _normalised(b['cond_cov'])
    for l in range(5):
        x = torch.complex(ca[l][..., 0], ca[l][..., 1]).numpy()
        y = torch.complex(cb[l][..., 0], cb[l][..., 1]).numpy()
        assert np.abs(y - x @ D[l].T).max() < 1e-5, (name, l)  # test_agent.py:59-61
    sa, sb = atomic_scalars_of(ca), atomic_scalars_of(cb)
    assert torch.allclose(sa, sb, atol=1e-5)  # test_agent.py:119
