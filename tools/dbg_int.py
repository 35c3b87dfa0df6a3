import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tests.test_gpu_internal import _pair, ZS, N
from molgym_amd.synthetic import make_batch_internal
ac, ref = _pair(0)
data = make_batch_internal(8, N, ZS, seed=4)
with torch.no_grad():
    out = ac.step(data['obs'], data['act'])
    exp = ref.step(data['obs'], data['act'], dtype=torch.float64)
nat=[sum(1 for it in o[0] if it[0]!=0) for o in data['obs']]
print('natoms',nat)
for k in ('logp','ent','v'):
    print(k, out[k].cpu().numpy().round(4), exp[k].numpy().round(4))
