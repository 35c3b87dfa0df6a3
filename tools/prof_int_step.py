import sys, cProfile, pstats, torch
sys.path.insert(0, '.')
from molgym_amd.agents.internal import SchNetAC
from molgym_amd.spaces import ActionSpace, ObservationSpace
from molgym_amd.synthetic import CONFIGS, make_batch
cfg = CONFIGS['cfg2']
obs = make_batch(140, cfg['canvas_size'], cfg['zs'], seed=0)['obs']
ia = SchNetAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']), (0.8, 1.8), 128, device='cuda:0')
ia.training = True
for _ in range(3): ia.step(obs)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): ia.step(obs)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(32)
