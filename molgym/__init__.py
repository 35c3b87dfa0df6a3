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
