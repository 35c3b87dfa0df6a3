"""The data-parallel path under RCCL on real devices: a short batch_ppo (per-rank environments, gather_rollout, sharded
train, broadcast permutation, flat-gradient all-reduce) launched through torchrun with one process per visible GPU
(a 1-GPU box runs world size 1 -- the same collectives, on the nccl backend).  The world-2 == world-1 equivalence of the
arithmetic is tested on the CPU over gloo (tests/test_dp_gloo.py)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batch_ppo_under_rccl(built_lib, tmp_path):
    n = max(1, min(torch.cuda.device_count(), 2))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'dp.pt')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'tests', 'dp_worker.py'), out]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    rec = torch.load(out)
    assert rec == {'moved': True, 'finite': True, 'replicas_equal': True, 'world': n}


def test_bench_rccl_path_forced_at_world_1(built_lib):
    """`bench.py --gpus 1 --force-dist`: the exact code the driver's 2 / 4 / 8-GPU scaling runs execute -- process group on the
    nccl (= RCCL) backend, barrier-bracketed timed region, max over ranks, the flat gradient all-reduced once per
    `--allreduce-every` steps, the replica check -- on the one GPU of this box, so that the GPU test record proves it runs."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--steps', '20', '--warmup', '5',
           '--no-cpu-baseline', '--no-epoch-overlap', '--no-build']
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1])
    d = line['dist']
    assert d['rccl_ranks'] == 1 and d['backend'] == 'nccl' and d['rank_devices'] == [0]
    assert d['replicas_equal'] is True and d['allreduces_in_timed_region'] == 2  # 20 steps, one all-reduce per 10
    assert line['n_gpus'] == 1 and line['steps'] == 20 and line['value'] > 0 and line['config']['allreduce_every_steps'] == 10
    assert line['config']['issued_as_one_graph_launch'] is True
