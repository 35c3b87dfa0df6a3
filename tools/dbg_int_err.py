"""Debug tool (GPU box): output errors of SchNetAC against the oracle for one (canvas, width); env switches choose the kernels."""
import sys
import torch
sys.path.insert(0, '.')
from tests.test_gpu_internal import _pair, ZS
from tests.helpers import rel_err
from molgym_amd.synthetic import make_batch_internal
canvas, width = int(sys.argv[1]), int(sys.argv[2])
for seed in (0, 1, 2):
    ac, ref = _pair(seed, width, canvas)
    data = make_batch_internal(24, canvas, ZS, seed=4 + seed)
    out = ac.step(data['obs'], data['act'])
    exp = ref.step(data['obs'], data['act'], dtype=torch.float64)
    print(seed, {k: float('%.3g' % rel_err(out[k].detach(), exp[k].detach())) for k in ('logp', 'ent', 'v')})
