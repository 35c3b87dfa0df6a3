"""PPO update driver on the HIP hot path.

Mirror of /root/reference/molgym/ppo.py:18-161 (compute_loss, get_batch_generator,
collect_data_batch, compute_mean_dict, train) with the same names, arguments and
return values.  `train` keeps the reference's semantics exactly -- gradients are
accumulated over every mini-batch of the rollout and ONE optimizer step is taken
per epoch (ppo.py:117-146) -- but each mini-batch runs as forward -> float64 loss
-> hand-written backward on the device with no host synchronisation.

Data parallel (new; the reference is single-process): when torch.distributed is
initialised every rank holds the same rollout `data`, draws the same permutation
(same numpy seed), takes its contiguous slice of every mini-batch, scales its
gradient by B_local / B_global and the flat gradient vector is summed ONCE per
epoch (one RCCL all-reduce of ~0.75 MB) before the norm / clip / step, so all
ranks take identical steps.
"""
import logging
import time
from typing import Dict, Iterator, List, Sequence, Tuple

import numpy as np
import torch

KEYS = ('policy_loss', 'entropy_loss', 'vf_loss', 'total_loss', 'approx_kl', 'clip_fraction')


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def compute_loss(ac, data: dict, clip_ratio: float, vf_coef: float, entropy_coef: float,
                 device=None) -> Tuple[torch.Tensor, Dict[str, float]]:
    """Reference-compatible form (ppo.py:18-63): autograd loss tensor (float64) + info dict."""
    pred = ac.step(data['obs'], data['act'])
    dev = pred['logp'].device
    old_logp = torch.as_tensor(data['logp'], device=dev)
    adv = torch.as_tensor(data['adv'], device=dev)
    ret = torch.as_tensor(data['ret'], device=dev)
    ratio = torch.exp(pred['logp'] - old_logp)
    obj = ratio * adv
    clipped_obj = ratio.clamp(1 - clip_ratio, 1 + clip_ratio) * adv
    policy_loss = -torch.min(obj, clipped_obj).mean()
    entropy_loss = -entropy_coef * pred['ent'].mean()
    vf_loss = vf_coef * (pred['v'] - ret).pow(2).mean()
    loss = policy_loss + entropy_loss + vf_loss
    approx_kl = (old_logp - pred['logp']).mean()
    clipped = ratio.lt(1 - clip_ratio) | ratio.gt(1 + clip_ratio)
    clip_fraction = torch.as_tensor(clipped, dtype=torch.float32).mean()
    vals = torch.stack([policy_loss, entropy_loss, vf_loss, loss, approx_kl, clip_fraction.double()]).tolist()
    return loss, dict(zip(KEYS, vals))


def get_batch_generator(indices: np.ndarray, batch_size: int) -> Iterator[np.ndarray]:
    assert len(indices.shape) == 1
    indices = np.random.permutation(indices)  # global numpy RNG, like the reference
    batches = indices[:len(indices) // batch_size * batch_size].reshape(-1, batch_size)
    for batch in batches:
        yield batch
    remainder = len(indices) % batch_size
    if remainder:
        yield indices[-remainder:]


def collect_data_batch(data: Dict[str, Sequence], indices: np.ndarray) -> Dict[str, Sequence]:
    batch: Dict[str, Sequence] = {}
    for key, value in data.items():
        if isinstance(value, np.ndarray):
            batch[key] = value[indices]
        elif isinstance(value, list):
            batch[key] = [value[i] for i in indices]
    return batch


def compute_mean_dict(dicts: List[Dict[str, float]]) -> Dict[str, float]:
    return {key: np.mean([d[key] for d in dicts]) for key in dicts[0].keys()}


def compute_gradient_norm(parameters) -> float:
    """util.compute_gradient_norm (tools/util.py:61-69): norm of the per-tensor norms."""
    ps = [p for p in parameters if p.grad is not None]
    if not ps:
        return 0.0
    return torch.norm(torch.stack([torch.norm(p.grad.detach(), 2) for p in ps]), 2).item()


def train(ac, optimizer, data: Dict[str, Sequence], mini_batch_size: int, clip_ratio: float, target_kl: float,
          vf_coef: float, entropy_coef: float, gradient_clip: float, max_num_steps: int, device=None) -> dict:
    infos: Dict[str, float] = {}
    start_time = time.time()
    dist, rank, world = _dist()
    fast = hasattr(ac, 'prepare_rollout')
    rollout = ac.prepare_rollout(data) if fast else None
    # The mini-batches of one epoch are independent given theta (their gradients only accumulate), and a small
    # mini-batch leaves most of the chip idle: keep up to 3 of them in flight on separate HIP streams.
    streams = []
    if fast and len(data['obs']) > mini_batch_size:
        streams = [torch.cuda.Stream(device=ac.theta.device) for _ in range(3)]
    num_epochs = 0
    for i in range(max_num_steps):
        optimizer.zero_grad()
        if fast and ac.theta.grad is None:
            ac.theta.grad = torch.zeros_like(ac.theta)
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        batch_stats, weights = [], []
        for mb_index, batch_indices in enumerate(get_batch_generator(np.arange(len(data['obs'])), mini_batch_size)):
            n_glob = len(batch_indices)
            lo, hi = (rank * n_glob) // world, ((rank + 1) * n_glob) // world
            local = batch_indices[lo:hi]
            if fast:
                if len(local):
                    mb = rollout.minibatch(local)
                    if streams:
                        slot = mb_index % len(streams)
                        streams[slot].wait_stream(torch.cuda.current_stream())  # the gather above ran there
                        with torch.cuda.stream(streams[slot]):
                            stats = ac.ppo_minibatch(mb, clip_ratio, vf_coef, entropy_coef,
                                                     loss_scale=len(local) / n_glob, slot=slot)
                            stats = stats * (len(local) / n_glob)
                        for t in (mb.pos, mb.charges, mb.bags, mb.actions, mb.logp, mb.adv, mb.ret, stats):
                            t.record_stream(streams[slot])
                        batch_stats.append(stats)
                    else:
                        stats = ac.ppo_minibatch(mb, clip_ratio, vf_coef, entropy_coef,
                                                 loss_scale=len(local) / n_glob)
                        batch_stats.append(stats * (len(local) / n_glob))
                else:
                    batch_stats.append(torch.zeros(6, dtype=torch.float64, device=ac.theta.device))
            else:
                loss, info = compute_loss(ac, collect_data_batch(data, local), clip_ratio, vf_coef, entropy_coef, device)
                (loss * (len(local) / n_glob)).backward()
                batch_stats.append(torch.tensor([info[k] for k in KEYS], dtype=torch.float64) * (len(local) / n_glob))
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        stats = torch.stack(batch_stats).mean(dim=0)  # mean of mini-batch means (ppo.py:92-95)
        if dist is not None:
            dist.all_reduce(stats)
            for p in ac.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad)
        loss_info = dict(zip(KEYS, stats.tolist()))
        loss_info['grad_norm'] = compute_gradient_norm(ac.parameters())
        if loss_info['approx_kl'] > 1.5 * target_kl:
            logging.debug(f'Early stopping at step {i} for reaching max KL.')
            break
        torch.nn.utils.clip_grad_norm_(ac.parameters(), max_norm=gradient_clip)
        optimizer.step()
        optimizer.zero_grad()
        num_epochs += 1
        infos.update(loss_info)
    infos['num_opt_steps'] = num_epochs
    infos['time'] = time.time() - start_time
    if num_epochs > 0:
        logging.info(f'Optimization: policy loss={infos["policy_loss"]:.3f}, vf loss={infos["vf_loss"]:.3f}, '
                     f'entropy loss={infos["entropy_loss"]:.3f}, total loss={infos["total_loss"]:.3f}, '
                     f'num steps={num_epochs}')
    return infos
