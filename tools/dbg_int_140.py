"""Debug tool (GPU box): per-sample errors of SchNetAC at B = 140 against the f64 oracle, beside the f32 oracle's own error."""
import sys
import torch
sys.path.insert(0, '.')
from tests.test_gpu_internal import _pair, ZS
from molgym_amd.synthetic import make_batch_internal
B = int(sys.argv[1]) if len(sys.argv) > 1 else 140
ac, ref = _pair(0, 128, 7)
data = make_batch_internal(B, 7, ZS, seed=4)
out = ac.step(data['obs'], data['act'])
exp = ref.step(data['obs'], data['act'], dtype=torch.float64)
import copy
ref32 = copy.deepcopy(ref).float()
e32 = ref32.step(data['obs'], data['act'], dtype=torch.float32)
for k in ('logp', 'ent', 'v'):
    got, want, w32 = out[k].detach().double().cpu(), exp[k].detach(), e32[k].detach().double()
    rel = (got - want).abs() / want.abs().clamp(min=1e-2)
    rel32 = (w32 - want).abs() / want.abs().clamp(min=1e-2)
    idx = torch.argsort(rel, descending=True)[:5]
    print(k, 'hip worst', [(int(i), float('%.4g' % want[i]), float('%.3g' % rel[i]), float('%.3g' % rel32[i])) for i in idx],
          'f32-oracle max', float('%.3g' % rel32.max()))
