"""Flat parameter vector of the covariant actor-critic.

theta is ONE float32 vector; a slot is a named tensor inside it.  Slot names are
the ``state_dict`` keys the reference module would have
(/root/reference/molgym/agents/covariant/agent.py:58-143 and
covariant/modules.py:52-95,155-178 give the attribute names), so
``export_state_dict`` / ``import_state_dict`` map 1:1 to a CovariantAC checkpoint.
The C side (molgym_amd/csrc/state.inc, build_layout) computes the same offsets;
tests cross-check them through mg_cov_param_offsets.
"""
from collections import OrderedDict
from typing import Tuple

MAXL, NLEV, CH, CE, NRADF = 4, 3, 10, 4, 32
NBLK = (5, 12, 16, 17, 15)


def edge_cin(k: int, l: int, ch: int = CH) -> int:
    return (2 * ch if l == 0 else ch) if k == 0 else 7 * ch


def atom_tau(k: int, l: int, ch: int = CH) -> int:
    return (3 * ch if l == 0 else ch) if k == 0 else ch * (2 * NBLK[l] + 1)


def mix_tau(l: int, ce: int = CE) -> int:
    return ce * (NBLK[l] + 2)


def slots(num_zs: int, width: int, num_gaussians: int, ch: int = CH, ce: int = CE, nlev: int = NLEV) -> 'OrderedDict[str, Tuple[int, ...]]':
    """`ch` / `ce`: num_channels_hidden / num_channels_per_element (arg_parser.py:55-60); the library build has to match
    (molgym_amd/_lib.py::lib(channels)); `nlev`: num_cg_levels (arg_parser.py:56), a build parameter of the library too"""
    CH, CE, NLEV = ch, ce, nlev  # noqa: N806 (shadow the defaults below)
    co = num_zs * CE
    nlat, nlat_e = (MAXL + 2) * co * 2, (MAXL + 2) * CE * 2
    s: 'OrderedDict[str, Tuple[int, ...]]' = OrderedDict()
    for k in range(NLEV):
        base = f'cg_model.rad_funcs.rad_funcs.{k}'
        s[f'{base}.scales'] = (1, 1, 1, 8)
        s[f'{base}.phases'] = (1, 1, 1, 8)
        for l in range(MAXL + 1):
            s[f'{base}.linear.{l}.weight'] = (2 * CH, NRADF)
            s[f'{base}.linear.{l}.bias'] = (2 * CH, )
    s['cg_model.input_func_atom.lin.weight'] = (2 * CH, 4 * num_zs)
    s['cg_model.input_func_atom.lin.bias'] = (2 * CH, )
    for k in range(NLEV):
        for l in range(MAXL + 1):
            s[f'cg_model.cormorant_cg.edge_levels.{k}.cat_mix.weights.{l}'] = (CH, edge_cin(k, l, CH), 2)
    for k in range(NLEV):
        cout = co if k == NLEV - 1 else CH
        for l in range(MAXL + 1):
            s[f'cg_model.cormorant_cg.atom_levels.{k}.cat_mix.weights.{l}'] = (cout, atom_tau(k, l, CH), 2)
    for l in range(MAXL + 1):
        s[f'cg_mix.cat_mix.weights.{l}'] = (CE, mix_tau(l, CE), 2)
    for name, n_in, n_out in (('phi_focus', nlat, 1), ('phi_element', nlat, num_zs),
                              ('phi_d', nlat_e, 2 * num_gaussians), ('phi_trans', nlat, width), ('phi_v', width, 1)):
        s[f'{name}.layers.0.weight'] = (width, n_in)
        s[f'{name}.layers.0.bias'] = (width, )
        s[f'{name}.layers.1.weight'] = (n_out, width)
        s[f'{name}.layers.1.bias'] = (n_out, )
    s['distance_log_stds'] = (num_gaussians, )
    return s


def offsets(num_zs: int, width: int, num_gaussians: int, ch: int = CH, ce: int = CE, nlev: int = NLEV):
    """name -> (offset, shape); also returns the total length."""
    out, off = OrderedDict(), 0
    for name, shape in slots(num_zs, width, num_gaussians, ch, ce, nlev).items():
        n = 1
        for d in shape:
            n *= d
        out[name] = (off, shape)
        off += n
    return out, off


# ---- internal-coordinate agent (SchNetAC) -----------------------------------------------------------------
SN_F, SN_G, SN_MAXZ, SN_T = 128, 25, 100, 3


def slots_internal(num_zs: int, width: int) -> 'OrderedDict[str, Tuple[int, ...]]':
    """state_dict-style names of /root/reference/molgym/agents/internal/agent.py:37-105 (the schnetpack 0.3
    SchNet module tree under `embedding_fn`; its filter_network is also aliased under cfconv there)."""
    af, lb = width // 2, width // 4
    nl = af + lb
    s: 'OrderedDict[str, Tuple[int, ...]]' = OrderedDict()
    s['embedding_fn.embedding.weight'] = (SN_MAXZ, af)
    for t in range(SN_T):
        b = f'embedding_fn.interactions.{t}'
        s[f'{b}.filter_network.0.weight'] = (SN_F, SN_G)
        s[f'{b}.filter_network.0.bias'] = (SN_F, )
        s[f'{b}.filter_network.1.weight'] = (SN_F, SN_F)
        s[f'{b}.filter_network.1.bias'] = (SN_F, )
        s[f'{b}.cfconv.in2f.weight'] = (SN_F, af)
        s[f'{b}.cfconv.f2out.weight'] = (af, SN_F)
        s[f'{b}.cfconv.f2out.bias'] = (af, )
        s[f'{b}.dense.weight'] = (af, af)
        s[f'{b}.dense.bias'] = (af, )
    for name, n_in, n_out in (('phi_beta', num_zs, lb), ('phi_focus', nl, 1), ('phi_element', nl, num_zs),
                              ('phi_continuous', nl + num_zs, 3), ('phi_kappa', nl, 1)):
        s[f'{name}.layers.0.weight'] = (width, n_in)
        s[f'{name}.layers.0.bias'] = (width, )
        s[f'{name}.layers.1.weight'] = (n_out, width)
        s[f'{name}.layers.1.bias'] = (n_out, )
    s['log_stds'] = (3, )
    s['critic.layers.0.weight'] = (width, nl)
    s['critic.layers.0.bias'] = (width, )
    s['critic.layers.1.weight'] = (width, width)
    s['critic.layers.1.bias'] = (width, )
    s['critic.layers.2.weight'] = (1, width)
    s['critic.layers.2.bias'] = (1, )
    return s


def offsets_internal(num_zs: int, width: int):
    out, off = OrderedDict(), 0
    for name, shape in slots_internal(num_zs, width).items():
        n = 1
        for d in shape:
            n *= d
        out[name] = (off, shape)
        off += n
    return out, off
