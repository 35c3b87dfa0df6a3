"""PPO driver on the HIP hot path.

Mirror of /root/reference/molgym/ppo.py (compute_loss :18-63, get_batch_generator :66-74, collect_data_batch :77-89,
compute_mean_dict :92-95, train :99-161, batch_rollout :164-218, compute_buffer_stats :221-227, batch_ppo :230-377)
with the same names, arguments and return values.

`train` keeps the reference's semantics exactly -- gradients are accumulated over every mini-batch of the rollout
and ONE optimizer step is taken per epoch (ppo.py:117-146) -- but for the HIP agents each mini-batch runs as
forward -> float64 loss -> hand-written backward on the device with no host synchronisation (`ppo_minibatch`), the
gradient norm / clip is one C call on the flat gradient (`mg_grad_norm_clip`), and the per-epoch host traffic is ONE
copy of 7 doubles.  An agent without that device path (anything with just `step`) goes through `compute_loss` +
autograd, mini-batch by mini-batch, like the reference.  Both ways share one sharding / reduction code path.

Data parallel (new; the reference is single-process).  When torch.distributed is initialised:
  * rollout: every rank steps its OWN environments; `gather_rollout` all-gathers the rollout as ONE float64 matrix per
    rank (observations as arrays, `ParsedObservations`; no pickling of Python tuples) in rank order and standardises the
    advantages over the merged buffer (buffer.py:104-110) -- exactly the `data` a single process driving all the
    environments would have built;
  * update: rank 0's permutation is broadcast each epoch (an int64 tensor); WHOLE mini-batches are dealt round-robin to the
    ranks (the reference accumulates the gradient over all mini-batches of an epoch and steps once, ppo.py:117-146, so
    which rank evaluates which mini-batch changes nothing, and B_local stays at the mini-batch size instead of
    mini_batch / world), the leftover mini-batches of an epoch -- fewer than `world` -- are sliced contiguously with
    gradient scale B_local / B_global (an empty slice contributes zero); the flat gradient is summed ONCE per epoch
    (one ~0.75 MB RCCL all-reduce) before norm / clip / step, so all ranks take identical Adam steps; the 6 loss
    statistics ride in one 48-byte float64 all-reduce.
  * `torchrun scripts/run.py` UNCHANGED: the `molgym` shim initialises the process group (molgym/__init__.py) and sets
    `DP_SHARD_GLOBAL_CONFIG`; `batch_ppo` then treats `envs` / `num_steps_per_iter` as the GLOBAL configuration every rank
    built identically, keeps this rank's share of the environments and offsets the rollout RNG streams by rank (the
    model was built before, from the same seed on every rank).
"""
import logging
import time
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .buffer import DynamicPPOBuffer, PPOBufferContainer
from .observations import ParsedObservations

KEYS = ('policy_loss', 'entropy_loss', 'vf_loss', 'total_loss', 'approx_kl', 'clip_fraction')
# set by the `molgym` shim when IT initialised torch.distributed for an unchanged reference script (see batch_ppo)
DP_SHARD_GLOBAL_CONFIG = False


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def to_numpy(t) -> np.ndarray:
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


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
        if isinstance(value, ParsedObservations):
            batch[key] = value.take(indices)
        elif isinstance(value, np.ndarray):
            batch[key] = value[indices]
        elif torch.is_tensor(value):  # get_data(device=...) hands over float64 device tensors
            batch[key] = value[torch.as_tensor(np.asarray(indices, dtype=np.int64), device=value.device)]
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


# ---- one epoch's mini-batches: device path / autograd path behind one interface ------------------------------------
class _DeviceRunner:
    """Agents with `prepare_rollout` + `ppo_minibatch` (CovariantAC, SchNetAC): the rollout is parsed once and parked
    in HBM, a mini-batch is a device gather + forward / loss / backward launches, statistics stay on the device.
    The mini-batches of one epoch are independent given theta (their gradients only accumulate) and a small
    mini-batch leaves most of the chip idle: up to 3 are kept in flight on separate HIP streams."""

    def __init__(self, ac, data, mini_batch_size, hp):
        self.ac, self.hp = ac, hp
        self.rollout = ac.prepare_rollout(data)
        self.num_samples, self._whole = len(data['obs']), None
        self.world = _dist()[2]
        self.dev = next(ac.parameters()).device
        self.streams = []
        if self.dev.type == 'cuda' and len(data['obs']) > mini_batch_size:
            import os
            nstreams = max(1, min(8, int(os.environ.get('MOLGYM_TRAIN_STREAMS', '3'))))  # (8 = the library's graph slots)
            self.streams = [torch.cuda.Stream(device=self.dev) for _ in range(nstreams)]
        # agents whose ppo_minibatch adds (share x statistics) into an epoch accumulator on the device (CovariantAC: inside the
        # loss kernel): no tensor per mini-batch, no `stats * scale` launch per mini-batch
        import inspect
        self._accumulates = 'stats_accum' in inspect.signature(ac.ppo_minibatch).parameters
        # agents that keep what depends on theta alone across the mini-batches of an epoch (CovariantAC: derived weights prepared once
        # per epoch and workspace, expanded weight gradients folded once per epoch): theta only changes between epochs here
        self._epoch_cache = 'epoch_cache' in inspect.signature(ac.ppo_minibatch).parameters and hasattr(ac, 'fold_gradients')
        self._acc = None

    def set_epoch(self, locals_: Sequence[np.ndarray]):
        """this rank's sample indices of every mini-batch of the epoch: ONE upload; `run` hands device views of it to the
        gather (a 140-sample mini-batch spent more host time on its own index upload + seven index_select calls than on
        the forward and backward launches)"""
        sizes = [len(x) for x in locals_]
        if self.world == 1 and sizes == [self.num_samples]:  # the whole rollout at once: gathered once (_minibatch)
            self._idx = None
            return
        flat = np.concatenate([np.asarray(x, dtype=np.int64) for x in locals_]) if sizes else np.zeros(0, dtype=np.int64)
        self._idx = torch.from_numpy(flat).to(self.dev) if self.dev.type == 'cuda' else None
        self._off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)

    def begin_epoch(self):
        for p in self.ac.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        if self._accumulates:
            self._acc = torch.zeros(6, dtype=torch.float64, device=self.dev)  # (on the current stream, ahead of the waits below)
        if self._epoch_cache:
            self.ac.invalidate_weights()  # the optimizer stepped since the last epoch
        for st in self.streams:
            st.wait_stream(torch.cuda.current_stream(self.dev))  # (also orders the index upload before the gathers)

    def _minibatch(self, mb_index: int, local: np.ndarray):
        if len(local) == self.num_samples and self.world == 1:
            # the mini-batch IS the rollout: a permutation of all samples changes neither the loss (a mean) nor the
            # gradient (a sum): one gather for all epochs
            if self._whole is None:
                self._whole = self.rollout.minibatch(np.arange(self.num_samples))
            return self._whole
        idx = getattr(self, '_idx', None)
        if idx is None:
            return self.rollout.minibatch(local)
        lo, hi = int(self._off[mb_index]), int(self._off[mb_index + 1])
        assert hi - lo == len(local)
        return self.rollout.minibatch(local, idx[lo:hi])

    def run(self, mb_index: int, local: np.ndarray, scale: float) -> torch.Tensor:
        if len(local) == 0:  # this rank's slice of a small remainder mini-batch
            return torch.zeros(6, dtype=torch.float64, device=self.dev)
        kw = {'stats_accum': self._acc} if self._accumulates else {}
        if self._epoch_cache:
            kw['epoch_cache'] = True
        if not self.streams:
            stats = self.ac.ppo_minibatch(self._minibatch(mb_index, local), *self.hp, loss_scale=scale, **kw)
            if self._accumulates:
                return None
            return stats if scale == 1.0 else stats * scale
        slot = mb_index % len(self.streams)
        st = self.streams[slot]
        with torch.cuda.stream(st):  # gather and step on the mini-batch's own stream: no cross-stream hand-off
            mb = self._minibatch(mb_index, local)
            stats = self.ac.ppo_minibatch(mb, *self.hp, loss_scale=scale, slot=slot, **kw)
            if not self._accumulates:
                stats = stats * scale
        if self._accumulates:
            return None
        stats.record_stream(torch.cuda.current_stream(self.dev))
        return stats

    def end_epoch(self):
        for st in self.streams:
            torch.cuda.current_stream(self.dev).wait_stream(st)
        if self._epoch_cache:
            self.ac.fold_gradients()  # theta.grad is complete from here on (all-reduce / norm / clip / Adam follow)

    def accumulated(self) -> Optional[torch.Tensor]:
        """sum over this rank's mini-batches of (share x statistics), where the agent accumulated it on the device"""
        return self._acc if self._accumulates else None


class _AutogradRunner:
    """Any AbstractActorCritic: compute_loss + loss.backward() per mini-batch (ppo.py:122-131)."""

    def set_epoch(self, locals_):
        pass

    def __init__(self, ac, data, mini_batch_size, hp, device=None):
        self.ac, self.data, self.hp, self.device = ac, data, hp, device
        self.dev = next(ac.parameters()).device

    def begin_epoch(self):
        pass

    def accumulated(self):
        return None

    def run(self, mb_index: int, local: np.ndarray, scale: float) -> torch.Tensor:
        if len(local) == 0:  # mean over zero samples is NaN: an empty slice contributes nothing
            return torch.zeros(6, dtype=torch.float64, device=self.dev)
        loss, info = compute_loss(self.ac, collect_data_batch(self.data, local), *self.hp, self.device)
        (loss * scale).backward()
        return torch.tensor([info[k] for k in KEYS], dtype=torch.float64, device=self.dev) * scale

    def end_epoch(self):
        pass


def _comm_device(dist) -> torch.device:
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')


class _SharedPermutations:
    """The epoch permutations of one `train` call under torch.distributed, WITHOUT a collective (and its device -> host
    synchronisation) per epoch.  The reference draws every epoch's permutation from the global numpy RNG (ppo.py:66-74) and
    the ranks' global streams differ (the rollout moved them apart), so the ranks must agree on rank 0's permutations.  Nothing
    else draws from that stream while `train` runs: ONE exchange of rank 0's generator state at the top of `train` lets every
    rank replay rank 0's draws from a private clone (the run-ahead loop has three mini-batches in flight -- a broadcast + `.cpu()`
    per epoch drained them at world > 1 only, on the path no single-GPU run covers).  Every rank still advances its own global
    stream by the same draws, as before."""

    def __init__(self, dist):
        name, keys, pos, has_gauss, cached = np.random.get_state()
        assert name == 'MT19937'
        packed = np.concatenate([keys.astype(np.int64), [pos, has_gauss], np.frombuffer(np.float64(cached).tobytes(), dtype=np.int64)])
        state = torch.from_numpy(packed).to(_comm_device(dist))
        dist.broadcast(state, src=0)
        got = state.cpu().numpy()
        self.rng = np.random.RandomState()
        self.rng.set_state(('MT19937', got[:624].astype(np.uint32), int(got[624]), int(got[625]),
                            float(np.frombuffer(got[626:627].tobytes(), dtype=np.float64)[0])))

    def get_state(self):
        return self.rng.get_state()

    def set_state(self, st):
        self.rng.set_state(st)


def _epoch_batches(num_samples: int, mini_batch_size: int, dist, rank: int, shared: Optional[_SharedPermutations] = None) -> List[np.ndarray]:
    batches = list(get_batch_generator(np.arange(num_samples), mini_batch_size))  # every rank advances its numpy RNG
    if dist is not None and num_samples > 0:  # ... but rank 0's permutation is the one everybody uses
        assert shared is not None
        flat, sizes = shared.rng.permutation(np.arange(num_samples)), [len(b) for b in batches]  # (rank 0's draw, replayed)
        if rank == 0:
            assert np.array_equal(flat[:len(batches[0])], batches[0])
        batches = [flat[lo:lo + n] for lo, n in zip(np.concatenate([[0], np.cumsum(sizes)[:-1]]), sizes)]
    return batches


def shard_epoch(batches: List[np.ndarray], rank: int, world: int) -> List[Tuple[np.ndarray, float]]:
    """This rank's work of one epoch: [(sample indices, share of their mini-batch)].  The first floor(M / world) * world
    mini-batches are dealt WHOLE, round-robin (share 1); the remaining < world ones are sliced contiguously over the ranks
    (share = B_local / B_global; possibly empty).  Summed over the ranks every mini-batch is covered exactly once."""
    whole = (len(batches) // world) * world
    work = [(b, 1.0) for k, b in enumerate(batches[:whole]) if k % world == rank]
    for b in batches[whole:]:
        n = len(b)
        lo, hi = (rank * n) // world, ((rank + 1) * n) // world
        work.append((b[lo:hi], (hi - lo) / n))
    return work


def _train_runahead(ac, optimizer, runner, num_samples: int, mini_batch_size: int, target_kl: float, gradient_clip: float,
                    max_num_steps: int, infos: dict) -> int:
    """The epochs of `train` WITHOUT a host round trip per epoch.  The reference reads the epoch's mean KL on the host, breaks
    before the optimizer step when it is over 1.5 x target_kl, clips and steps otherwise (ppo.py:133-146) -- a synchronisation
    per epoch that drains the three mini-batches in flight and leaves the GPU idle until the next epoch's first launches arrive
    (~0.6 ms of a 3.7 ms epoch on the SF6 rollout).  Here the test, the norm and the clip run on the device
    (`ac.ppo_epoch_end`), the Adam launch is gated by the latching stop flag, and the host issues epoch i + 1 BEFORE it reads
    epoch i's record: the GPU always has an epoch queued.  Same result as the synchronous loop: once an epoch's KL is over the
    limit nothing updates theta any more; the one epoch issued speculatively behind it is discarded (its optimizer-step counter
    is taken back, the numpy RNG is put back to where the reference's loop would have left it)."""
    dist, rank, world = _dist()
    dev = runner.dev
    on_gpu = dev.type == 'cuda'
    shared = _SharedPermutations(dist) if dist is not None else None
    stop = torch.zeros(1, dtype=torch.int32, device=dev)
    recs_dev = torch.zeros(max_num_steps, 8, dtype=torch.float64, device=dev)
    recs_host = torch.zeros(max_num_steps, 8, dtype=torch.float64)
    if on_gpu:
        recs_host = recs_host.pin_memory()
    events, rng_states, keep = [], [], []
    flags, issued = None, 0
    for p in ac.parameters():
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    for i in range(max_num_steps):
        rng_states.append(np.random.get_state())
        for p in ac.parameters():
            p.grad.zero_()  # (optimizer.zero_grad(), in place: the launches of the epoch in flight hold the tensor's address)
        batches = _epoch_batches(num_samples, mini_batch_size, dist, rank, shared)
        slices = shard_epoch(batches, rank, world)
        runner.set_epoch([sl for sl, _ in slices])
        runner.begin_epoch()
        for mb_index, (sl, share) in enumerate(slices):
            runner.run(mb_index, sl, share)
        runner.end_epoch()
        acc = runner.accumulated()
        keep.append(acc)  # (written by the mini-batches' own streams: alive until the loop has drained)
        if dist is not None:
            dist.all_reduce(acc)
            for p in ac.parameters():
                dist.all_reduce(p.grad)
        ac.ppo_epoch_end(gradient_clip, acc, len(batches), 1.5 * target_kl, recs_dev[i], stop)
        ac.adam_step(optimizer, skip_flag=stop)
        recs_host[i].copy_(recs_dev[i], non_blocking=True)
        if i == 0 and hasattr(ac, 'input_flags_async'):
            flags = ac.input_flags_async()  # inconsistent inputs raise at the first look below instead of training on garbage
        # the event goes on the AGENT's stream (the one the record's copy was issued on): with an agent on cuda:1 while cuda:0 is
        # current, a bare `record()` lands on cuda:0's stream and `synchronize()` returns before the copy has
        ev = torch.cuda.Event() if on_gpu else None
        if on_gpu:
            with torch.cuda.device(dev):
                ev.record(torch.cuda.current_stream(dev))
        events.append(ev)
        issued += 1
        if i >= 1:  # epoch i is queued: now look at epoch i - 1
            if on_gpu:
                events[i - 1].synchronize()
            if flags is not None:
                ac.raise_input_flags(flags)
                flags = None
            if recs_host[i - 1, 7].item() != 0.0:
                break
    if events and on_gpu:
        events[-1].synchronize()
    if flags is not None:
        ac.raise_input_flags(flags)
    stopped = [k for k in range(issued) if recs_host[k, 7].item() != 0.0]
    num_epochs = stopped[0] if stopped else issued
    ac.adam_unstep(optimizer, issued - num_epochs)
    if stopped:
        logging.debug(f'Early stopping at step {num_epochs} for reaching max KL.')
        if issued > num_epochs + 1:  # a speculative epoch drew a permutation the reference's loop never draws
            np.random.set_state(rng_states[num_epochs + 1])
    if num_epochs > 0:
        row = recs_host[num_epochs - 1].tolist()
        infos.update(dict(zip(KEYS, row[:6])))
        infos['grad_norm'] = row[6]
    optimizer.zero_grad()
    return num_epochs


def train(ac, optimizer, data: Dict[str, Sequence], mini_batch_size: int, clip_ratio: float, target_kl: float,
          vf_coef: float, entropy_coef: float, gradient_clip: float, max_num_steps: int, device=None) -> dict:
    infos: Dict[str, float] = {}
    start_time = time.time()
    dist, rank, world = _dist()
    hp = (clip_ratio, vf_coef, entropy_coef)
    device_path = hasattr(ac, 'prepare_rollout') and hasattr(ac, 'ppo_minibatch')
    runner = _DeviceRunner(ac, data, mini_batch_size, hp) if device_path else \
        _AutogradRunner(ac, data, mini_batch_size, hp, device)
    # (`flat_gradient_on_host`: a stand-in agent that implements the flat-gradient calls in torch on the CPU -- tests/test_dp_gloo.py)
    flat = hasattr(ac, 'grad_norm_clip') and (next(ac.parameters()).device.type == 'cuda' or
                                               getattr(ac, 'flat_gradient_on_host', False))
    num_samples = len(data['obs'])
    num_epochs = 0
    import os
    runahead = device_path and flat and getattr(runner, '_accumulates', False) and hasattr(ac, 'ppo_epoch_end') and \
        max_num_steps >= 2 and num_samples > 0 and os.environ.get('MOLGYM_RUNAHEAD', '1') != '0'
    if runahead:
        for p in ac.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        runahead = ac.adam_supported(optimizer)
    if runahead:
        num_epochs = _train_runahead(ac, optimizer, runner, num_samples, mini_batch_size, target_kl, gradient_clip,
                                     max_num_steps, infos)
        max_num_steps = 0  # (the synchronous loop below is skipped)
    shared = _SharedPermutations(dist) if dist is not None and max_num_steps > 0 else None
    for i in range(max_num_steps):
        optimizer.zero_grad()
        batches = _epoch_batches(num_samples, mini_batch_size, dist, rank, shared)
        slices = shard_epoch(batches, rank, world)  # (this rank's sample indices, their share of the mini-batch)
        runner.set_epoch([sl for sl, _ in slices])
        runner.begin_epoch()
        batch_stats = [runner.run(mb_index, sl, share) for mb_index, (sl, share) in enumerate(slices)]
        runner.end_epoch()
        # mean of mini-batch means (ppo.py:92-95): every entry is (share x the mean over its samples); over all ranks the
        # entries of one mini-batch add up to its mean
        batch_stats = [st for st in batch_stats if st is not None]
        stats = torch.zeros(6, dtype=torch.float64, device=runner.dev)
        if batch_stats:
            stats = stats + torch.stack(batch_stats).sum(dim=0)
        if runner.accumulated() is not None:
            stats = stats + runner.accumulated()
        stats = stats / max(len(batches), 1)
        if dist is not None:
            dist.all_reduce(stats)
            for p in ac.parameters():
                if p.grad is None:  # a rank whose every slice was empty still takes part in the reduction
                    p.grad = torch.zeros_like(p)
                dist.all_reduce(p.grad)
        if flat:
            # norm AND clip in one call, ahead of the KL test: the norm reported is the pre-clip one, and when the test stops
            # the loop the (clipped) gradient is discarded anyway (ppo.py:135-144 breaks before the step)
            host = torch.cat([stats, ac.grad_norm_clip(gradient_clip).double()]).tolist()  # the epoch's only device -> host copy
            loss_info = dict(zip(KEYS, host[:6]))
            loss_info['grad_norm'] = host[6]
            if i == 0 and hasattr(ac, 'check_inputs'):
                ac.check_inputs()  # the stream is idle right here: inconsistent inputs raise instead of training on garbage
        else:
            loss_info = dict(zip(KEYS, stats.tolist()))
            loss_info['grad_norm'] = compute_gradient_norm(ac.parameters())
        if loss_info['approx_kl'] > 1.5 * target_kl:
            logging.debug(f'Early stopping at step {i} for reaching max KL.')
            break
        if not flat:
            torch.nn.utils.clip_grad_norm_(ac.parameters(), max_norm=gradient_clip)
        if not (flat and hasattr(ac, 'adam_step') and ac.adam_step(optimizer)):  # one launch for Adam on the flat theta
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


# ---- rollout ---------------------------------------------------------------------------------------------------------
def _host_predictions(predictions: dict) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(a, v, logp) of one step() on the host with ONE device -> host copy."""
    a, v, logp = predictions['a'], predictions['v'], predictions['logp']
    if torch.is_tensor(a) and a.is_cuda:
        packed = torch.cat([a.detach().float(), v.detach().float().unsqueeze(1), logp.detach().float().unsqueeze(1)],
                           dim=1).cpu().numpy()
        width = a.shape[1]
        return packed[:, :width], packed[:, width], packed[:, width + 1]
    return to_numpy(a), to_numpy(v), to_numpy(logp)


def batch_rollout(ac, envs, buffer_container: PPOBufferContainer, num_steps: Optional[int] = None,
                  num_episodes: Optional[int] = None, pipeline: int = 2) -> dict:
    """ppo.py:164-218.  With an `AsyncEnvContainer` (anything with `groups`) and a fixed number of steps the
    environments are split into `pipeline` groups living on disjoint worker processes and the loop is software-
    pipelined: while the workers step group g on the host (reward = 3 PM6 single points per environment), the GPU
    evaluates the policy for the next group.  Environments are independent, so every environment sees the same
    sequence of (observation -> action -> next observation) pairs either way."""
    assert num_steps is not None or num_episodes is not None
    start_time = time.time()
    if num_steps is not None and num_episodes is None and pipeline > 1 and hasattr(envs, 'groups') \
            and len(envs.groups(pipeline)) > 1:
        _rollout_pipelined(ac, envs, buffer_container, num_steps, pipeline)
    else:
        _rollout_serial(ac, envs, buffer_container, num_steps, num_episodes)
    return {
        'time': time.time() - start_time,
        'return_mean': np.mean(buffer_container.episodic_returns).item(),
        'return_std': np.std(buffer_container.episodic_returns).item(),
        'episode_length_mean': np.mean(buffer_container.episode_lengths).item(),
        'episode_length_std': np.std(buffer_container.episode_lengths).item(),
    }


def _use_canvas(ac) -> bool:
    return hasattr(ac, 'make_canvas') and next(ac.parameters()).device.type == 'cuda'


def _rollout_serial(ac, envs, container, num_steps, num_episodes):
    if num_steps is not None:
        assert num_steps % envs.get_size() == 0
        num_iters = num_steps // envs.get_size()
    else:
        num_iters = np.inf
    if num_episodes is not None:
        assert envs.get_size() == 1
    else:
        num_episodes = np.inf
    counter = 0
    observations = envs.reset()
    # agents with device-resident canvases (CovariantAC on the GPU): parse once, then only reset environments are uploaded
    canvas = ac.make_canvas(observations) if _use_canvas(ac) else None
    while counter < num_iters and container.get_num_episodes() < num_episodes:
        predictions = ac.step_canvas(canvas) if canvas is not None else ac.step(observations)
        next_observations, rewards, terminals, _ = envs.step(predictions['actions'])
        a, v, logp = _host_predictions(predictions)
        container.store(observations=observations, actions=a, rewards=rewards, next_observations=next_observations,
                        terminals=terminals, values=v, logps=logp)
        observations = envs.reset_if_terminal(next_observations, terminals)  # a valid next observation either way
        if canvas is not None:
            stale = canvas.stale_rows(observations, terminals)
            canvas.sync(stale, [observations[i] for i in stale])
        if counter == num_iters - 1:
            last = ac.step_canvas(canvas, commit=False) if canvas is not None else ac.step(observations)
            _, v, _ = _host_predictions(last)
            container.finish_paths(v)  # bootstrap the cut-off paths; finished ones are untouched
        counter += 1


def _rollout_pipelined(ac, envs, container, num_steps, pipeline):
    assert num_steps % envs.get_size() == 0
    num_iters = num_steps // envs.get_size()
    groups = envs.groups(pipeline)
    obs = [envs.reset(g) for g in groups]
    # one set of device-resident canvases PER GROUP (CovariantAC on the GPU): a group's policy evaluation is then a sampling
    # launch on resident arrays -- no parse, no upload -- while the other group's environments step on the host
    canvases = [ac.make_canvas(o) if _use_canvas(ac) else None for o in obs]
    pending = [None] * len(groups)  # (ticket, host predictions) of the step in flight per group
    # Sampling: the serial rollout draws ONE seed per step and row b of the batch reads the random stream (seed, b).  Here the
    # groups of one step share that step's seed and key their rows by ENVIRONMENT id (groups are g, g + k, g + 2k, ...: an
    # affine map), so every environment sees the same draws as in the serial rollout and the buffers come out identical.
    affine = all(list(g) == list(range(g[0], g[0] + len(groups) * len(g), len(groups))) for g in groups) and \
        sorted(g[0] for g in groups) == list(range(len(groups)))
    seeds: List[int] = []

    def seed_of(step):  # seeds are drawn in step order, one per step, like the serial loop (num_iters + 1 in total)
        while len(seeds) <= step:
            seeds.append(ac.draw_seed())
        return seeds[step]

    def evaluate(k, step, commit=True):
        if canvases[k] is not None:
            if affine:
                return ac.step_canvas(canvases[k], commit=commit, seed=seed_of(step), sample_ids=(groups[k][0], len(groups)))
            return ac.step_canvas(canvases[k], commit=commit)
        return ac.step(obs[k])

    def launch(k, step):
        predictions = evaluate(k, step)
        pending[k] = (envs.step_async(predictions['actions'], groups[k]), _host_predictions(predictions))

    for k in range(len(groups)):
        launch(k, 0)  # group k+1's policy evaluation overlaps group k's environment step
    for it in range(num_iters):
        for k, g in enumerate(groups):
            ticket, (a, v, logp) = pending[k]
            next_obs, rewards, terminals, _ = envs.step_wait(ticket)
            container.store(observations=obs[k], actions=a, rewards=rewards, next_observations=next_obs,
                            terminals=terminals, values=v, logps=logp, env_indices=g)
            obs[k] = envs.reset_if_terminal(next_obs, terminals, g)
            if canvases[k] is not None:
                stale = canvases[k].stale_rows(obs[k], terminals)
                canvases[k].sync(stale, [obs[k][i] for i in stale])
            if it < num_iters - 1:
                launch(k, it + 1)
            else:
                _, v, _ = _host_predictions(evaluate(k, num_iters, commit=False))
                container.finish_paths(v, env_indices=g)


def compute_buffer_stats(buffer: DynamicPPOBuffer) -> Dict[str, float]:
    return {
        'value_mean': np.mean(buffer.val_buf).item(),
        'value_std': np.std(buffer.val_buf).item(),
        'logp_mean': np.mean(buffer.logp_buf).item(),
        'logp_std': np.std(buffer.logp_buf).item(),
    }


def all_gather_rows(mat: np.ndarray) -> np.ndarray:
    """Rows of a float64 matrix of every rank, concatenated in rank order (ranks may hold different numbers of rows): one
    all-gather of the counts, one of the zero-padded matrices."""
    dist, rank, world = _dist()
    assert dist is not None and mat.ndim == 2
    dev = _comm_device(dist)
    count = torch.tensor([mat.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    padded = torch.zeros(max(max(counts), 1), mat.shape[1], dtype=torch.float64, device=dev)
    padded[:mat.shape[0]] = torch.from_numpy(np.ascontiguousarray(mat, dtype=np.float64)).to(dev)
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return np.concatenate([p[:n].cpu().numpy() for p, n in zip(parts, counts)], axis=0)


def gather_rollout(buffer: DynamicPPOBuffer) -> dict:
    """`buffer.get_data()` of the rollout of ALL ranks (every rank stepped its own environments), merged in rank order --
    the buffer one process driving all the environments would have built -- with the advantages standardised over the
    merged buffer (buffer.py:104-110).  The exchange is tensor collectives on arrays: observations travel as the
    [labels | positions | bag] matrix of `ParsedObservations` next to act / ret / adv / logp / val, ONE padded float64
    all-gather per iteration (the earlier all_gather_object pickled every observation tuple of the rollout on every rank).
    `data['obs']` is a `ParsedObservations` (a Sequence of the reference's tuples; the HIP agents' parsers take its arrays
    directly); `data['val']` (extra key, unused by `train`) lets rank 0 log buffer statistics of the whole rollout.
    Without torch.distributed this is `buffer.get_data()`."""
    dist, rank, world = _dist()
    if dist is None:
        return buffer.get_data()
    assert buffer.is_finished()
    T = len(buffer.obs_buf)
    shape = torch.tensor([0, 0, 0], dtype=torch.int64, device=_comm_device(dist))  # canvas size, labels, action width
    if T:
        obs = ParsedObservations.from_list(buffer.obs_buf)
        act = np.asarray(buffer.act_buf, dtype=np.float64).reshape(T, -1)
        shape = torch.tensor([obs.canvas_size, obs.num_labels, act.shape[1]], dtype=torch.int64, device=shape.device)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX)  # a rank with an empty buffer still has to size its (empty) matrix
    N, Z, A = (int(v) for v in shape.tolist())
    if T:
        local = np.concatenate([obs.to_matrix(), act, np.asarray(buffer.ret_buf, dtype=np.float64)[:, None],
                                np.asarray(buffer.adv_buf, dtype=np.float64)[:, None],
                                np.asarray(buffer.logp_buf, dtype=np.float64)[:, None],
                                np.asarray(buffer.val_buf, dtype=np.float64)[:, None]], axis=1)
    else:
        local = np.zeros((0, 4 * N + Z + A + 4), dtype=np.float64)
    full = all_gather_rows(local)
    k = 4 * N + Z
    adv = full[:, k + A + 1]
    return dict(obs=ParsedObservations.from_matrix(full[:, :k], N, Z), act=full[:, k:k + A], ret=full[:, k + A],
                adv=(adv - np.mean(adv)) / np.std(adv), logp=full[:, k + A + 2], val=full[:, k + A + 3])


def _global_rollout_info(container: PPOBufferContainer, info: dict) -> dict:
    """Episode statistics of the rollouts of ALL ranks (rank 0 logs and saves them): episodic returns / lengths gathered
    as one small matrix."""
    dist, rank, world = _dist()
    if dist is None:
        return info
    rows = np.stack([np.asarray(container.episodic_returns, dtype=np.float64),
                     np.asarray(container.episode_lengths, dtype=np.float64)], axis=1).reshape(-1, 2)
    full = all_gather_rows(rows)
    out = dict(info)
    if len(full):
        out.update(return_mean=np.mean(full[:, 0]).item(), return_std=np.std(full[:, 0]).item(),
                   episode_length_mean=np.mean(full[:, 1]).item(), episode_length_std=np.std(full[:, 1]).item())
    return out


def _shard_global_config(envs, num_steps_per_iter: int, rank: int, world: int):
    """`torchrun scripts/run.py` unchanged: every rank built ALL `num_envs` environments and was told the GLOBAL
    `num_steps_per_iter`.  Keep this rank's contiguous share of the environments and of the steps (so the job as a
    whole is the single-process configuration), and move the rollout RNG streams apart (numpy: environments /
    stochastic bags; torch: action sampling) -- the model exists already, built from the same seed on every rank.
    Applies only when the `molgym` shim itself brought the process group up for a script that knows nothing about ranks
    (DP_SHARD_GLOBAL_CONFIG; a rank-aware script initialises torch.distributed itself -- or sets MOLGYM_NO_DIST=1 -- and
    passes per-rank environments / steps, which are then taken as they are).  The container is marked, so a second
    batch_ppo call in the same process does not shard the already-sharded environments again."""
    mark = getattr(envs, '_mg_dp_shard', None)
    if mark is not None and mark[:2] == (rank, world):
        # a second batch_ppo call on the same container (curriculum stage, resume): the environments ARE this rank's share
        # already and the RNG streams were moved apart then -- only the (again global) step count is converted
        size = mark[2]
    else:
        size = envs.get_size()
        if not hasattr(envs, 'environments'):
            raise RuntimeError('data-parallel launch of an unchanged script: the training container must expose `.environments`')
        if size < world:
            raise RuntimeError(f'data-parallel launch of an unchanged script: {size} environments cannot be dealt to {world} ranks')
    if num_steps_per_iter % size:
        raise RuntimeError(f'num_steps_per_iter ({num_steps_per_iter}) must be a multiple of num_envs ({size}) '
                           '(ppo.py:175: every environment takes the same number of steps per iteration)')
    if mark is None or mark[:2] != (rank, world):
        # contiguous shares that differ by at most one environment: every environment belongs to exactly one rank, also when
        # num_envs is not a multiple of the world size
        lo, hi = (rank * size) // world, ((rank + 1) * size) // world
        envs.environments = envs.environments[lo:hi]
        envs._mg_dp_shard = (rank, world, size)
        base = int(np.random.randint(2**31 - world))  # the same draw on every rank (same seed so far)
        np.random.seed(base + rank)
        torch.manual_seed(base + rank)
    return envs, (num_steps_per_iter // size) * envs.get_size()


def batch_ppo(
    envs,
    eval_envs,
    ac,
    optimizer,
    gamma=0.99,
    start_num_steps=0,
    max_num_steps=4096,
    num_steps_per_iter=200,
    mini_batch_size=64,
    clip_ratio=0.2,
    vf_coef=0.5,
    entropy_coef=0.0,
    max_num_train_iters=80,
    lam=0.97,
    target_kl=0.01,
    gradient_clip=0.5,
    save_freq=5,
    model_handler=None,
    eval_freq=10,
    num_eval_episodes=1,
    rollout_saver=None,
    save_train_rollout=False,
    save_eval_rollout=True,
    info_saver=None,
    device=None,
):
    """PPO-clip with KL early stopping: the reference's main loop (ppo.py:230-377), argument for argument.
    Under torch.distributed `envs` are THIS rank's environments and `num_steps_per_iter` counts this rank's steps;
    rank 0 alone evaluates, logs and saves."""
    dist, rank, world = _dist()
    steps_per_iter_global = num_steps_per_iter * world  # per-rank environments / steps handed in by a rank-aware caller
    if dist is not None and DP_SHARD_GLOBAL_CONFIG:
        steps_per_iter_global = num_steps_per_iter      # the script's own (global) configuration
        envs, num_steps_per_iter = _shard_global_config(envs, num_steps_per_iter, rank, world)
    total_num_steps = start_num_steps
    num_iterations = (max_num_steps - total_num_steps) // steps_per_iter_global
    logging.info('Starting PPO')
    for iteration in range(num_iterations):
        logging.info(f'Iteration: {iteration}/{num_iterations - 1}, steps: {total_num_steps}')
        train_container = PPOBufferContainer(size=envs.get_size(), gamma=gamma, lam=lam)
        train_rollout = batch_rollout(ac=ac, envs=envs, buffer_container=train_container,
                                      num_steps=num_steps_per_iter)
        logging.info(f'Training rollout: return={train_rollout["return_mean"]:.3f} '
                     f'({train_rollout["return_std"]:.1f}), episode length={train_rollout["episode_length_mean"]:.1f}')
        train_buffer = train_container.merge()
        data = None
        if dist is not None:  # rank 0 logs the rollout of ALL ranks
            data = gather_rollout(train_buffer)
            train_rollout = _global_rollout_info(train_container, train_rollout)
        if info_saver and rank == 0:
            train_rollout['total_num_steps'] = total_num_steps
            if data is not None:
                train_rollout.update(value_mean=np.mean(data['val']).item(), value_std=np.std(data['val']).item(),
                                     logp_mean=np.mean(data['logp']).item(), logp_std=np.std(data['logp']).item())
            else:
                train_rollout.update(compute_buffer_stats(train_buffer))
            info_saver.save(train_rollout, name='train')
        if rollout_saver and save_train_rollout:  # every rank saves ITS buffer; ranks > 0 tag the file
            rollout_saver.save(train_buffer, num_steps=total_num_steps, info='train' if rank == 0 else f'train-rank{rank}')

        if data is not None:
            pass
        elif hasattr(ac, 'prepare_rollout') and next(ac.parameters()).device.type == 'cuda':
            data = train_buffer.get_data(device=next(ac.parameters()).device)  # GAE + standardisation on the device
        else:
            data = train_buffer.get_data()
        opt_info = train(ac=ac, optimizer=optimizer, data=data, mini_batch_size=mini_batch_size,
                         clip_ratio=clip_ratio, vf_coef=vf_coef, entropy_coef=entropy_coef, target_kl=target_kl,
                         gradient_clip=gradient_clip, max_num_steps=max_num_train_iters, device=device)
        if info_saver and rank == 0:
            opt_info['total_num_steps'] = total_num_steps
            info_saver.save(opt_info, name='opt')
        total_num_steps += steps_per_iter_global

        if rank == 0 and ((iteration % eval_freq == 0) or (iteration == num_iterations - 1)):
            eval_container = PPOBufferContainer(size=eval_envs.get_size(), gamma=gamma, lam=lam)
            with torch.no_grad():
                ac.training = False  # the reference flips the attribute directly (ppo.py:353,361)
                eval_rollout = batch_rollout(ac, eval_envs, buffer_container=eval_container,
                                             num_episodes=num_eval_episodes)
                logging.info(f'Evaluation rollout: return={eval_rollout["return_mean"]:.3f} '
                             f'({eval_rollout["return_std"]:.1f}), '
                             f'episode length={eval_rollout["episode_length_mean"]:.1f}')
                ac.training = True
            eval_buffer = eval_container.merge()
            if info_saver:
                eval_rollout['total_num_steps'] = total_num_steps
                eval_rollout.update(compute_buffer_stats(eval_buffer))
                info_saver.save(eval_rollout, name='eval')
            if rollout_saver and save_eval_rollout:
                rollout_saver.save(eval_buffer, num_steps=total_num_steps, info='eval')
        if rank == 0 and model_handler and ((iteration % save_freq == 0) or (iteration == num_iterations - 1)):
            model_handler.save(ac, num_steps=total_num_steps)
        if dist is not None:
            dist.barrier()
    logging.info('Finished PPO')
