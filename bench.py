#!/usr/bin/env python3
"""PPO mini-batch forward+backward throughput of the covariant actor-critic on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic mini-batch (molgym/ppo.py:124-131):
``step(obs, actions)`` -> float64 PPO loss -> backward into ``.grad`` (optimizer step excluded), with
the parsed mini-batch already resident in HBM.  N > 1: one process per GPU -- launched by torchrun, or by this
script itself when it is started plainly with --gpus N -- with the flat gradient all-reduced over RCCL inside the
timed region -- ONCE per `--allreduce-every` steps (default 10: the mini-batches of one ppo.train epoch accumulate
locally and the epoch ends in one all-reduce, ppo.py:117-146 / molgym_amd/ppo.py::train; the line states the cadence; the
gradient buffer is zeroed on the same cadence at every N, `config.zero_grad_every_steps`);
`--scaling weak` (default) gives every rank its own mini-batch of the configured size every step, `--scaling strong`
deals the K WHOLE mini-batches round-robin to the ranks, the way `ppo.train` shards an epoch (total work fixed).
Rank 0 prints ONE JSON line; `value` follows the contract
(K steps between synchronisations, max over ranks), `config.median_ms_per_step` is the median over the same K steps
from per-step HIP events.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3  # MI355X f32 vector == f32-input MFMA dense peak (MI355X_MICROARCH.md)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=50)
    p.add_argument('--warmup', type=int, default=10)
    p.add_argument('--config', default='cfg2', help='synthetic workload (molgym_amd/synthetic.py)')
    p.add_argument('--batch', type=int, default=None, help='override the mini-batch size per GPU')
    p.add_argument('--agent', default='covariant', choices=['covariant', 'internal'],
                   help="'internal' = SchNetAC (BASELINE configs[0]); single GPU, no roofline object")
    p.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                   help='N > 1: weak = the configured mini-batch PER GPU; strong = ONE mini-batch of that size sharded '
                        'over the GPUs (what ppo.train does with a fixed rollout)')
    p.add_argument('--allreduce-every', type=int, default=10,
                   help='N > 1: all-reduce the accumulated gradient once per this many steps (one ppo.train epoch)')
    p.add_argument('--inflight', type=int, default=1, help='mini-batches in flight per GPU (independent HIP streams)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-epoch-overlap', action='store_true', help='skip the extra legs (three in flight; host parse inside): profiling runs')
    p.add_argument('--no-build', action='store_true', help='use the library as is (A/B runs with MOLGYM_HIP_LIB)')
    p.add_argument('--cpu-seconds', type=float, default=20.0, help='budget of the CPU baseline leg')
    p.add_argument('--force-dist', action='store_true',
                   help='exercise the multi-GPU code path at N = 1 too: re-exec under torchrun, RCCL process group, '
                        'gradient all-reduce in the step (a 1-GPU box can then check what the 8-GPU run will execute)')
    return p.parse_args()


def _cpu_baseline_worker(cfg_name, B, seed, budget_s, threads, sd_path):
    """Runs in a child process (bounded by a timeout in the parent)."""
    import torch
    from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch
    from oracle.covariant_ref import CovariantACRef
    from oracle.ppo_ref import compute_loss_ref
    cfg = CONFIGS[cfg_name]
    torch.set_num_threads(threads)
    ref = CovariantACRef(zs=cfg['zs'], canvas_size=cfg['canvas_size'], bag_scale=cfg['bag_scale'], beta=cfg['beta'],
                         **MODEL_DEFAULTS)
    ref.load_state_dict(torch.load(sd_path))
    data = make_batch(B, cfg['canvas_size'], cfg['zs'], seed=seed)

    def one():
        for p in ref.parameters():
            p.grad = None
        loss, _ = compute_loss_ref(ref, data, 0.2, 0.5, 0.01)
        loss.backward()

    one()  # warm-up
    t0, n = time.perf_counter(), 0
    while True:
        one()
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 20:
            break
    print(json.dumps({'dt': (time.perf_counter() - t0) / n, 'n': n}))


def cpu_baseline(cfg_name, B, seed, state_dict, budget_s):
    """Reference-equivalent CPU restatement (oracle/) timed on the host cores: checker/baseline only.
    The GPU box has hundreds of cores and this path is thousands of tiny ATen ops, where an OpenMP team
    of os.cpu_count() threads is pathologically slow -- a few team sizes are tried, the best is reported."""
    import subprocess
    import tempfile
    import torch
    cores = os.cpu_count() or 1
    with tempfile.TemporaryDirectory() as tmp:
        sd_path = os.path.join(tmp, 'sd.pt')
        torch.save(state_dict, sd_path)
        best = None
        cands = sorted({min(cores, t) for t in (8, 32)})
        for threads in cands:
            code = ('import sys; sys.path.insert(0, %r); import bench; '
                    'bench._cpu_baseline_worker(%r, %d, %d, %f, %d, %r)' %
                    (ROOT, cfg_name, B, seed, budget_s / len(cands), threads, sd_path))
            try:
                res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True,
                                     timeout=4 * budget_s + 120)
                rec = json.loads(res.stdout.strip().splitlines()[-1])
            except Exception:  # timed out / failed: skip this team size
                continue
            if best is None or rec['dt'] < best[0]:
                best = (rec['dt'], rec['n'], threads)
    if best is None:
        return None
    dt, n, threads = best
    return {'value': B / dt, 'unit': 'samples/s', 'cores': threads, 'kind': 'port',
            'sample': f'{n} fwd+bwd passes over the same {B}-sample mini-batch, float32, torch {torch.__version__}, '
                      f'{threads} threads (best of {cands}; host has {cores} cores), oracle/ restatement at the '
                      f'reference op granularity'}


def _cpu_baseline_worker_internal(B, seed, budget_s, threads, sd_path):
    """SchNetAC leg of the CPU baseline (child process)."""
    import torch
    from molgym_amd.synthetic import make_batch_internal
    from oracle.internal_ref import SchNetACRef
    from oracle.ppo_ref import compute_loss_ref
    torch.set_num_threads(threads)
    ref = SchNetACRef([0, 9, 16], 7, (0.8, 1.8), 128)
    ref.load_state_dict(torch.load(sd_path), strict=True)
    data = make_batch_internal(B, 7, [0, 9, 16], seed=seed)
    t0, n = time.perf_counter(), 0
    while True:
        ref.zero_grad()
        loss, _ = compute_loss_ref(ref, data, 0.2, 0.5, 0.01)
        loss.backward()
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 10:
            break
    print(json.dumps({'dt': (time.perf_counter() - t0) / n, 'n': n}))


def main_internal(args):
    """`--agent internal`: SchNetAC forward + float64 loss + backward on BASELINE configs[0] (SF6, canvas 7,
    mini-batch 140), the ragged 3B-molecule batch resident in HBM.  Extra measurement, not the driver's metric."""
    import subprocess
    import tempfile
    import numpy as np
    import torch
    import __graft_entry__ as entry
    if not args.no_build:
        entry.build()
    from molgym_amd.agents.internal import SchNetAC
    from molgym_amd.spaces import ActionSpace, ObservationSpace
    from molgym_amd.synthetic import make_batch_internal
    zs, N = [0, 9, 16], 7
    B = args.batch or 140
    torch.manual_seed(0)
    ac = SchNetAC(ObservationSpace(N, zs), ActionSpace(zs), (0.8, 1.8), 128, device='cuda:0')
    data = make_batch_internal(B, N, zs, seed=0)
    batch = ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret'])
    ac.theta.grad = torch.zeros_like(ac.theta)

    every, count = max(1, args.allreduce_every), [0]

    epoch_cache = os.environ.get('MG_EPOCH_CACHE', '1') != '0'

    def step():
        if count[0] % every == 0:  # once per epoch of `every` mini-batches (see the covariant leg)
            ac.theta.grad.zero_()
            ac.invalidate_weights()  # the optimizer steps once per epoch: the derived weights are prepared by its first mini-batch
        count[0] += 1
        return ac.ppo_minibatch(batch, 0.2, 0.5, 0.01, epoch_cache=epoch_cache)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if not torch.isfinite(stats).all():
        raise SystemExit('non-finite loss statistics')
    line = {'metric': 'PPO mini-batch fwd+bwd samples/sec (internal, canvas_size=7)', 'value': B * args.steps / elapsed,
            'unit': 'samples/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic', 'timed_region_ms': elapsed * 1e3,
            'config': {'workload': f'configs[0]: SchNet internal-coordinate actor-critic, zs={zs}, canvas_size={N}, '
                                   f'mini_batch={B}, 3B-molecule ragged batch resident in HBM'},
            'roofline': None, 'cpu_baseline': None}
    from molgym_amd.profile import internal_roofline
    line['roofline'] = internal_roofline(ac, batch)
    if not args.no_cpu_baseline:
        with tempfile.TemporaryDirectory() as tmp:
            sd_path = os.path.join(tmp, 'sd.pt')
            torch.save({k: v.float().cpu() for k, v in ac.export_state_dict().items()}, sd_path)
            code = ('import sys; sys.path.insert(0, %r); import bench; '
                    'bench._cpu_baseline_worker_internal(%d, 0, %f, 8, %r)' % (ROOT, B, args.cpu_seconds, sd_path))
            try:
                res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True,
                                     timeout=4 * args.cpu_seconds + 120)
                rec = json.loads(res.stdout.strip().splitlines()[-1])
                line['cpu_baseline'] = {'value': B / rec['dt'], 'unit': 'samples/s', 'cores': 8, 'kind': 'port',
                                        'sample': f"{rec['n']} fwd+bwd passes over the same {B}-sample mini-batch, "
                                                  f'float32, 8 threads, oracle/ restatement'}
            except Exception:
                pass
    print(json.dumps(line))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def main():
    args = parse_args()
    if args.agent == 'internal':
        return main_internal(args)
    if (args.gpus > 1 or args.force_dist) and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` on its own: become the torchrun launch the driver would have made
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    wd = float(os.environ.get('BENCH_WATCHDOG', '0'))
    if wd > 0:  # dump all Python stacks and exit if the run wedges
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    import numpy as np
    import torch
    import __graft_entry__ as entry
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)
    if rank == 0 and not args.no_build:
        entry.build()
    if use_dist:
        dist.barrier()

    from molgym_amd.agents.covariant import CovariantAC
    from molgym_amd.spaces import ActionSpace, ObservationSpace
    from molgym_amd.synthetic import CONFIGS, MODEL_DEFAULTS, make_batch
    from tools.flops import forward_flops, step_flops

    cfg = dict(CONFIGS[args.config])
    B_glob = args.batch or cfg['batch']
    if args.scaling == 'weak':   # every rank owns a mini-batch of the configured size
        B = B_glob
        data = make_batch(B, cfg['canvas_size'], cfg['zs'], seed=rank)
    else:                        # strong: the K whole mini-batches dealt round-robin to the ranks (ppo.shard_epoch)
        B = B_glob
        data = make_batch(B, cfg['canvas_size'], cfg['zs'], seed=0)
    total_samples = world * B if args.scaling == 'weak' else B_glob  # samples per timed step over all ranks
    torch.manual_seed(0)
    ac = CovariantAC(ObservationSpace(cfg['canvas_size'], cfg['zs']), ActionSpace(cfg['zs']),
                     bag_scale=cfg['bag_scale'], beta=cfg['beta'], device=dev, **MODEL_DEFAULTS)
    batch = ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret'])
    ac.theta.grad = torch.zeros_like(ac.theta)
    # weak: a step is `world` mini-batches, the all-reduced gradient their mean; strong: whole mini-batches, scale 1
    loss_scale = 1.0 / world if args.scaling == 'weak' else 1.0
    every = max(1, args.allreduce_every)
    steps_asked = args.steps
    if use_dist and args.scaling == 'strong':
        # strong scaling deals WHOLE mini-batches round-robin: with K = 20 and 8 ranks a rank would time 2-3 mini-batches
        # (about 1 ms).  K is rounded up to a multiple of world x allreduce_every: a whole number of epochs of `every` mini-batches
        # in which every rank evaluates exactly K / world of them; the line reports both numbers
        quantum = world * every
        args.steps = -(-args.steps // quantum) * quantum
        args.warmup = -(-max(args.warmup, 1) // world) * world

    # MG_EPOCH_CACHE=0: derived weights and the fold of the expanded weight gradients per mini-batch (the round-4 step; A/B)
    epoch_cache = os.environ.get('MG_EPOCH_CACHE', '1') != '0'
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.inflight)] if args.inflight > 1 else None
    counter = [0]

    last_stats = [torch.zeros(6, dtype=torch.float64, device=dev)]
    n_allreduce = [0]

    def step(i=0, last=False):
        if streams is None:
            if i % every == 0:
                # once per epoch of `every` mini-batches, at any N: the reference zeroes the gradient per EPOCH and lets the
                # epoch's mini-batches accumulate (ppo.py:117-131).  (Through round 2 the one-GPU run zeroed before every step
                # -- one extra 4 us launch per step that the product does not make and that the N > 1 runs never had.)
                ac.theta.grad.zero_()
                # ... and steps the optimizer once per epoch (ppo.py:145): what depends on theta alone (the derived weight matrices)
                # is prepared by the epoch's first mini-batch and reused by the others, as ppo.train does (epoch_cache)
                ac.invalidate_weights()
            mine = args.scaling == 'weak' or i % world == rank
            if mine:
                last_stats[0] = ac.ppo_minibatch(batch, 0.2, 0.5, 0.01, loss_scale=loss_scale, epoch_cache=epoch_cache)
            if (i + 1) % every == 0 or last:
                ac.fold_gradients()  # the epoch's expanded complex weight gradients -> theta.grad (once per epoch, like the all-reduce)
                if use_dist:
                    dist.all_reduce(ac.theta.grad)  # one flat f32 bucket over RCCL / xGMI per epoch
                    n_allreduce[0] += 1
            return last_stats[0]
        # epoch semantics of ppo.train: gradients of independent mini-batches accumulate; they are issued
        # round-robin on `inflight` streams with their own workspaces
        k = counter[0] % len(streams)
        if counter[0] % every == 0:
            ac.invalidate_weights()
        counter[0] += 1
        with torch.cuda.stream(streams[k]):
            return ac.ppo_minibatch(batch, 0.2, 0.5, 0.01, loss_scale=loss_scale, slot=k, epoch_cache=epoch_cache)

    def drain():
        if streams is not None:
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
            ac.fold_gradients()
            if use_dist:
                dist.all_reduce(ac.theta.grad)

    for i in range(args.warmup):
        step(i, i == args.warmup - 1)
    drain()
    n_allreduce_warm = [n_allreduce[0]]
    # per-step HIP events on the launch stream (no synchronisation inside the timed region): median step time
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        stats = step(i, i == args.steps - 1)
        if streams is None:
            marks[i + 1].record()
    drain()
    t_issued = time.perf_counter() - t0  # host time to enqueue the K steps (diagnostic: CPU-bound if ~= elapsed)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if not torch.isfinite(stats).all():
        raise SystemExit('non-finite loss statistics')
    step_used_graph = bool(getattr(ac, 'last_step_used_graph', False))  # (the live roofline leg below runs eagerly: events per kernel)
    step_launches = int(ac._L().mg_cov_step_launches()) or None
    dist_info = None
    if use_dist:
        # evidence for the scaling record: how many RCCL ranks there were, which device each drove, and that after the last
        # all-reduce of the timed region every replica holds the SAME gradient (bit for bit: RCCL's ring sum is the same
        # sequence of additions on every rank)
        devs = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
        props = torch.cuda.get_device_properties(dev)
        # (the physical identity too: a launcher that narrows the visible devices per rank makes every rank's index 0)
        phys = (int(getattr(props, 'pci_domain_id', 0)) << 16) | (int(getattr(props, 'pci_bus_id', -1)) << 8) | int(getattr(props, 'pci_device_id', 0))
        dist.all_gather(devs, torch.tensor([torch.cuda.current_device(), phys], dtype=torch.int64, device=dev))
        g = ac.theta.grad
        probe = torch.stack([g.double().sum(), g.double().abs().sum(), g[::997].double().pow(2).sum()])
        lo, hi = probe.clone(), probe.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist_info = {'rccl_ranks': dist.get_world_size(), 'backend': dist.get_backend(),
                     'rank_devices': [int(d[0].item()) for d in devs],
                     'rank_pci': [int(d[1].item()) for d in devs],
                     'replicas_equal': bool(torch.equal(lo, hi)) and bool(torch.isfinite(probe).all()),
                     'grad_abs_sum': float(probe[1].item()),
                     'allreduces_in_timed_region': n_allreduce[0] - n_allreduce_warm[0]}
        # the 8-GPU run must fail loudly or pass, never quietly degrade (a rank that fell back to another device, a process
        # group of the wrong size, replicas that drifted apart): checked on every rank, non-zero exit
        problems = []
        if dist_info['rccl_ranks'] != args.gpus:
            problems.append(f"process group has {dist_info['rccl_ranks']} ranks, --gpus {args.gpus}")
        if dist_info['backend'] != 'nccl':
            problems.append(f"backend {dist_info['backend']} is not RCCL")
        if len(set(dist_info['rank_pci'])) != world and len(set(dist_info['rank_devices'])) != world:
            problems.append(f"ranks share devices: {dist_info['rank_devices']} / pci {dist_info['rank_pci']}")
        if not dist_info['replicas_equal']:
            problems.append('replicas hold different gradients after the last all-reduce')
        if dist_info['allreduces_in_timed_region'] < 1:
            problems.append('no gradient all-reduce inside the timed region')
        if problems:
            if rank == 0:
                print(json.dumps({'error': 'multi-GPU check failed', 'problems': problems, 'dist': dist_info}))
            dist.destroy_process_group()
            raise SystemExit(3)
    # Second leg (one GPU, default run only): the same K mini-batch steps with three in flight on separate HIP streams, the
    # way ppo.train issues the mini-batches of ONE epoch -- the reference zeroes the gradient once per epoch and lets the
    # mini-batches accumulate into it (molgym/ppo.py:117-131), so they are independent given theta.  Reported beside the
    # headline (which stays strictly one mini-batch at a time), never instead of it.
    epoch_leg = None
    if streams is None and not use_dist and world == 1 and B <= 512 and not args.no_epoch_overlap:
        ep_streams = [torch.cuda.Stream(device=dev) for _ in range(3)]

        def ep_step(i):
            if i % every == 0:
                ac.invalidate_weights()
            with torch.cuda.stream(ep_streams[i % 3]):
                return ac.ppo_minibatch(batch, 0.2, 0.5, 0.01, loss_scale=loss_scale, slot=i % 3, epoch_cache=epoch_cache)

        def ep_drain():
            for st in ep_streams:
                torch.cuda.current_stream().wait_stream(st)
            ac.fold_gradients()

        for st in ep_streams:
            st.wait_stream(torch.cuda.current_stream())
        for i in range(max(3, args.warmup)):
            ep_step(i)
        ep_drain()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            ep_stats = ep_step(i)
        ep_drain()
        torch.cuda.synchronize()
        ep_elapsed = time.perf_counter() - t1
        if not torch.isfinite(ep_stats).all():
            raise SystemExit('non-finite loss statistics (epoch leg)')
        epoch_leg = {'minibatches_in_flight': 3, 'steps': args.steps, 'value': total_samples * args.steps / ep_elapsed,
                     'unit': 'samples/s', 'ms_per_step': ep_elapsed / args.steps * 1e3,
                     'note': 'same mini-batch steps issued round-robin on 3 HIP streams (own workspaces), gradients '
                             'accumulating in one buffer: the mini-batches of one ppo.train epoch'}
    # Third leg (one GPU): the reference's timed region also holds the observation parsing of every mini-batch
    # (agent.py:165-197 inside step(), ppo.py:124-131): the same K steps with `prepare_batch` (host parse + upload) inside
    parse_leg = None
    if streams is None and not use_dist and world == 1 and not args.no_epoch_overlap:
        for _ in range(2):
            ac.theta.grad.zero_()
            ac.ppo_minibatch(ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret']), 0.2, 0.5, 0.01)
        torch.cuda.synchronize()
        k_parse = max(1, min(args.steps, 20 if B > 512 else args.steps))
        t2 = time.perf_counter()
        for it in range(k_parse):
            if it % every == 0:
                ac.theta.grad.zero_()
            ac.ppo_minibatch(ac.prepare_batch(data['obs'], data['act'], data['logp'], data['adv'], data['ret']), 0.2, 0.5, 0.01)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t2) / k_parse
        parse_leg = {'value': B / dt, 'unit': 'samples/s', 'ms_per_step': dt * 1e3, 'steps': k_parse,
                     'note': 'the same step with the host-side observation parse + upload of the mini-batch inside the '
                             'timed region (vectorised parser, ONE packed host -> device copy for positions / charges / bags / actions)'}
    median_ms = None
    if streams is None:
        median_ms = float(np.median([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]))

    # Fourth / fifth legs (one GPU, default run): (a) the SAME step on the schedule of the reference's README command for SF6
    # (/root/reference/README.md:79-80: --num_steps_per_iter=140 --mini_batch_size=140 => ONE mini-batch per epoch, theta changes
    # after every mini-batch): gradient zero, derived-weight preparation and the fold of the expanded weight gradients on EVERY
    # step, nothing amortised; (b) a long run of the headline's schedule (>= 2000 steps where the step is short), mean and median,
    # so that the driver's 20-step sample can be cross-read.
    def timed_leg(k, cadence):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        for i in range(min(10, k)):  # (re-warm: the graph execs of this cadence)
            one_step(i, cadence, i == min(10, k) - 1)
        torch.cuda.synchronize()
        t = time.perf_counter()
        ev[0].record()
        for i in range(k):
            st_ = one_step(i, cadence, i == k - 1)
            ev[i + 1].record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        if not torch.isfinite(st_).all():
            raise SystemExit('non-finite loss statistics (extra leg)')
        per = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(k)])
        return {'steps': k, 'value': total_samples * k / dt, 'unit': 'samples/s', 'ms_per_step': dt / k * 1e3,
                'median_ms_per_step': float(np.median(per)), 'p10_ms': float(np.percentile(per, 10)),
                'p90_ms': float(np.percentile(per, 90)), 'timed_region_ms': dt * 1e3}

    def one_step(i, cadence, last):
        if i % cadence == 0:
            ac.theta.grad.zero_()
            ac.invalidate_weights()
        out_ = ac.ppo_minibatch(batch, 0.2, 0.5, 0.01, loss_scale=loss_scale, epoch_cache=epoch_cache)
        if (i + 1) % cadence == 0 or last:
            ac.fold_gradients()
        return out_

    cadence1_leg = long_leg = None
    if streams is None and not use_dist and world == 1 and not args.no_epoch_overlap:
        ms_now = elapsed / args.steps * 1e3
        cadence1_leg = timed_leg(max(args.steps, min(200, int(200.0 / ms_now) + 1)), 1)
        cadence1_leg['note'] = ('every step is an epoch (the reference README SF6 command: one 140-sample mini-batch per epoch): '
                                'gradient zero + derived weights + weight-gradient fold on EVERY step')
        long_leg = timed_leg(max(args.steps, min(2000, int(1500.0 / ms_now) + 1)), every)
        long_leg['note'] = f"the headline's schedule (zero / derived weights / fold every {every} steps), long sample"

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = total_samples * args.steps / elapsed
        natoms = [sum(1 for it in o[0] if cfg['zs'][it[0]] != 0) for o in data['obs']]
        f_dense = 3 * forward_flops(cfg['canvas_size'], len(cfg['zs']))  # SURVEY 8(d) per-sample figure
        f_ragged = step_flops(natoms, len(cfg['zs'])) / B
        # dominant kernel, timed live with HIP events on the launch stream
        from molgym_amd.profile import dominant_kernel_roofline
        roof = dominant_kernel_roofline(ac, batch, natoms, cfg, args.config)
        line = {
            'metric': 'PPO mini-batch fwd+bwd samples/sec (covariant, canvas_size=%d)' % cfg['canvas_size'],
            'value': value, 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms, 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            # wall time between the two synchronisations that bracket the K timed steps (max over ranks): a thin sample is
            # visible at a glance
            'timed_region_ms': elapsed * 1e3,
            'config': {'workload': f'{args.config}: covariant actor-critic, zs={cfg["zs"]}, canvas_size='
                                   f'{cfg["canvas_size"]}, mini_batch={B} on this rank ({total_samples} over '
                                   f'{world} GPU(s), {args.scaling} scaling), beta={cfg["beta"]}, random-walk '
                                   f'canvases with U{{0..N}} atoms, inputs resident in HBM',
                       'global_batch': total_samples, 'parallelism': f'dp{world}' + (' (RCCL path forced)' if args.force_dist and world == 1 else ''),
                       'minibatches_in_flight': args.inflight,
                       'median_ms_per_step': median_ms,
                       'samples_per_s_at_median': None if median_ms is None else total_samples / (median_ms * 1e-3),
                       'host_enqueue_ms_per_step': t_issued / args.steps * 1e3,
                       # how the step reaches the GPU: mg_cov_ppo_step records its kernel launches and issues them as ONE
                       # hipGraph launch whose nodes are updated in place (include/molgym_hip.h); MG_GRAPH=0: plain stream launches
                       'issued_as_one_graph_launch': step_used_graph,
                       'kernel_launches_per_step': step_launches,
                       'step_tflops_dense_convention': f_dense * value / 1e12,
                       'step_tflops_ragged': f_ragged * value / 1e12,
                       'frac_f32_peak_dense_convention': f_dense * value / 1e12 / (PEAK_F32_TFLOPS * world),
                       # the same with the flops of the REAL atoms only (the kernels skip the padding the dense count includes)
                       'step_frac_ragged': f_ragged * value / 1e12 / (PEAK_F32_TFLOPS * world),
                       'allreduce_every_steps': every if use_dist else None,
                       'zero_grad_every_steps': every,
                       # per epoch of `every` mini-batches, not per mini-batch (theta is constant over an epoch, ppo.py:117-146):
                       # the derived weight matrices are re-derived and the expanded complex weight gradients folded into the
                       # gradient at this cadence -- INSIDE the timed region, like the zero and the all-reduce
                       'derived_weights_every_steps': every if epoch_cache else 1,
                       'gradient_fold_every_steps': every if epoch_cache else 1,
                       'steps_requested': steps_asked,
                       'steps_note': None if steps_asked == args.steps else
                       f'--steps {steps_asked} rounded up to {args.steps} = a multiple of world x allreduce_every '
                       f'({world} x {every}): strong scaling deals whole mini-batches round-robin, every rank times '
                       f'{args.steps // world} of them'},
            'dist': dist_info,
            'roofline': roof,
            'epoch_overlap': epoch_leg,
            'with_host_parse': parse_leg,
            'every_step_is_an_epoch': cadence1_leg,
            'long_run': long_leg,
        }
        line['config']['value_every_step_is_an_epoch'] = None if cadence1_leg is None else cadence1_leg['value']
        line['config']['value_long_run'] = None if long_leg is None else long_leg['value']
        from molgym_amd.profile import step_hbm
        line['config'].update(step_hbm(args.config, ms))
        if not args.no_cpu_baseline and world == 1:
            sd = {k: v.float().cpu() for k, v in ac.export_state_dict().items()}
            line['cpu_baseline'] = cpu_baseline(args.config, B, rank, sd, args.cpu_seconds)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
