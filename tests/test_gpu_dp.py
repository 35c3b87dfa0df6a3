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


def test_bench_rccl_path_forced_at_world_1_strong_scaling(built_lib):
    """the `--scaling strong` form of the same run (whole mini-batches dealt round-robin like ppo.shard_epoch; K rounded up
    to a multiple of world x allreduce_every): what a fixed-rollout 8-GPU run executes."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--scaling', 'strong', '--steps', '20',
           '--warmup', '5', '--no-cpu-baseline', '--no-epoch-overlap', '--no-build']
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1])
    d = line['dist']
    assert line['scaling'] == 'strong' and d['rccl_ranks'] == 1 and d['replicas_equal'] is True
    assert d['allreduces_in_timed_region'] == 2 and line['steps'] == 20 and line['value'] > 0


def test_bench_line_of_the_internal_agent(built_lib):
    """BASELINE configs[0] (SchNetAC, SF6, canvas 7, mini-batch 140): `bench.py --agent internal` produces a line with the
    `roofline` and `cpu_baseline` objects (a short CPU leg), so the driver's GPU record proves that measurement path."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--agent', 'internal', '--steps', '20', '--warmup', '5', '--no-build',
           '--cpu-seconds', '3']
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1])
    assert 'internal' in line['metric'] and line['value'] > 0 and line['steps'] == 20
    assert line['roofline'] and line['roofline'].get('frac') is not None
    assert line['cpu_baseline'] and line['cpu_baseline']['value'] > 0 and line['cpu_baseline']['kind'] == 'port'
