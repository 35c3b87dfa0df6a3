"""`molgym` shim: lets the reference's scripts (scripts/run.py:6-13 import only `molgym.*`) run UNCHANGED on the
MI355X hot path.

Put this repository's root on PYTHONPATH ahead of the reference checkout:

    MOLGYM_REFERENCE=/path/to/molgym-checkout  PYTHONPATH=/path/to/this/repo  python scripts/run.py ...

`import molgym.<x>` is then resolved over three directories, in this order:
  1. molgym/_hip/       the hot path, re-exported from `molgym_amd`: `molgym.ppo` (train, batch_ppo, batch_rollout,
                        compute_loss, ...), `molgym.agents.*` (CovariantAC, SchNetAC, AbstractActorCritic),
                        `molgym.tools.model_util` (build_model, ModelIO), `molgym.buffer`, `molgym.buffer_container`,
                        `molgym.env_container` (+ AsyncEnvContainer);
  2. <reference>/molgym the reference's own modules for everything that is NOT on the hot path -- environment,
                        reward, calculator, spaces, tools.util, tools.arg_parser, ... -- found through the
                        MOLGYM_REFERENCE environment variable (a checkout root) or any later sys.path entry that
                        holds a `molgym/environment.py`;
  3. molgym/_fallback/  gym- and ase-free stand-ins (`molgym.spaces`, `molgym.tools.util`) used only when no
                        reference checkout is reachable (this repo's tests, the GPU box).
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def _reference_package_dir():
    roots = [os.environ['MOLGYM_REFERENCE']] if os.environ.get('MOLGYM_REFERENCE') else []
    roots += [p for p in sys.path if p]
    for root in roots:
        cand = os.path.join(root, 'molgym')
        if os.path.abspath(cand) != _HERE and os.path.isfile(os.path.join(cand, 'environment.py')):
            return cand
    return None


REFERENCE_DIR = _reference_package_dir()


def search_path(*sub):
    """package search path for the (sub)package molgym[.sub...]: hot path first, reference second, fallback last"""
    dirs = [os.path.join(_HERE, '_hip', *sub)]
    if REFERENCE_DIR:
        dirs.append(os.path.join(REFERENCE_DIR, *sub))
    dirs.append(os.path.join(_HERE, '_fallback', *sub))
    return [d for d in dirs if os.path.isdir(d)]


__path__ = search_path()


def _init_distributed_from_env():
    """`python -m torch.distributed.run --nproc-per-node N scripts/run.py ...` with the script UNCHANGED: the reference's
    scripts know nothing about process groups (scripts/run.py:23-60) and `util.init_device` hands every rank plain
    `torch.device('cuda')` (tools/util.py:186-194).  When the launcher's environment is there (RANK / WORLD_SIZE /
    LOCAL_RANK, world > 1) and nobody initialised torch.distributed yet, this import does it: the rank's GPU becomes the
    current device (so that 'cuda' IS this rank's GPU), the process group comes up on RCCL ('nccl'; 'gloo' without a GPU),
    and `molgym.ppo.batch_ppo` is told to treat the script's arguments as the GLOBAL configuration (it keeps this rank's
    share of the environments and of the steps, offsets the rollout RNG streams by rank; the model, built from the common
    seed before, is identical everywhere).  MOLGYM_NO_DIST=1 switches this off; MOLGYM_DIST_BACKEND overrides the backend."""
    if os.environ.get('MOLGYM_NO_DIST') == '1':
        return
    try:
        world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '-1'))
    except ValueError:
        return
    if world <= 1 or rank < 0:
        return
    import torch
    import torch.distributed as dist
    if not dist.is_available() or dist.is_initialized():
        return
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    use_gpu = torch.cuda.is_available()
    backend = os.environ.get('MOLGYM_DIST_BACKEND') or ('nccl' if use_gpu else 'gloo')
    if use_gpu:
        torch.cuda.set_device(local % torch.cuda.device_count())
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', torch.cuda.current_device()))
    else:
        dist.init_process_group(backend)
    from molgym_amd import ppo as _ppo
    _ppo.DP_SHARD_GLOBAL_CONFIG = True


_init_distributed_from_env()
