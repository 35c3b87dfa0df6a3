import sys, torch, numpy as np
sys.path.insert(0, '.')
from tests.helpers import make_pair
from molgym_amd.synthetic import make_batch
lv = int(sys.argv[1])
ac, ref, cfg = make_pair('cfg2', seed=27, num_cg_levels=lv)
data = make_batch(12, cfg['canvas_size'], cfg['zs'], seed=35)
out = ac.step(data['obs'], data['act'])
torch.cuda.synchronize()
print('forward ok', flush=True)
(out['logp'].double().sum() + out['v'].double().sum()).backward()
torch.cuda.synchronize()
print('backward ok', flush=True)
