"""debug: accumulated gradient + statistics of one PPO epoch (mini-batches 16, 16, 8 on three streams) -> file; compare two library builds"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from molgym_amd import ppo
from molgym_amd.synthetic import make_batch
from tests.helpers import make_pair
out = sys.argv[1]
ac, ref, cfg = make_pair('cfg2', seed=22)
data = make_batch(40, cfg['canvas_size'], cfg['zs'], seed=10)
res = {}
for trial in range(3):
    ac.theta.grad = torch.zeros_like(ac.theta)
    ac.invalidate_weights()
    stats = []
    for a, b in ((0, 16), (16, 32), (32, 40)):
        bt = ac.prepare_batch(data['obs'][a:b], data['act'][a:b], data['logp'][a:b], data['adv'][a:b], data['ret'][a:b])
        stats.append(ac.ppo_minibatch(bt, 0.2, 0.5, 0.01, loss_scale=1.0, epoch_cache=True).clone())
    ac.fold_gradients()
    torch.cuda.synchronize()
    res[f'g{trial}'] = ac.theta.grad.cpu().numpy().copy()
    res[f's{trial}'] = torch.stack(stats).cpu().numpy()
np.savez(out, **res)
g = res['g0']
print('grad absmax', np.abs(g).max(), 'run-to-run', np.abs(res['g1'] - g).max(), np.abs(res['g2'] - g).max(), 'stats', res['s0'][0])
