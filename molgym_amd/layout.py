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


def cg_blocks(l: int, maxl: int = MAXL):
    """(l1, l2) blocks of output degree l of a CG product of two maxl-limited SO(3) vectors, l1 outer / l2 inner (cormorant's
    channel order; csrc/gen_tables.py::blocks): 5, 12, 16, 17, 15 blocks for maxl = 4"""
    return [(l1, l2) for l1 in range(maxl + 1) for l2 in range(maxl + 1) if abs(l1 - l2) <= l <= min(l1 + l2, maxl)]


def nblk(l: int, maxl: int = MAXL) -> int:
    return len(cg_blocks(l, maxl))


def edge_cin(k: int, l: int, ch: int = CH, maxl: int = MAXL) -> int:
    """[previous edge net | DotMatrix block, one ch per degree | radial] from level 1 on"""
    return (2 * ch if l == 0 else ch) if k == 0 else (maxl + 3) * ch


def atom_tau(k: int, l: int, ch: int = CH, maxl: int = MAXL) -> int:
    return (3 * ch if l == 0 else ch) if k == 0 else ch * (2 * nblk(l, maxl) + 1)


def mix_tau(l: int, ce: int = CE, maxl: int = MAXL) -> int:
    return ce * (nblk(l, maxl) + 2)


def slots(num_zs: int, width: int, num_gaussians: int, ch: int = CH, ce: int = CE, nlev: int = NLEV,
          maxl: int = MAXL) -> 'OrderedDict[str, Tuple[int, ...]]':
    """`ch` / `ce`: num_channels_hidden / num_channels_per_element (arg_parser.py:55-60); the library build has to match
    (molgym_amd/_lib.py::lib(channels)); `nlev`: num_cg_levels (arg_parser.py:56), a build parameter of the library too;
    `maxl` (arg_parser.py:56): the kernels are laid out for 4 -- a smaller maxl is the SAME network with the degrees above it
    structurally absent, and CovariantAC embeds its parameters into the maxl = 4 layout (`embedding_index`)"""
    CH, CE, NLEV, MAXL = ch, ce, nlev, maxl  # noqa: N806 (shadow the defaults below)
    edge_cin_ = lambda k, l, c: edge_cin(k, l, c, maxl)  # noqa: E731
    atom_tau_ = lambda k, l, c: atom_tau(k, l, c, maxl)  # noqa: E731
    mix_tau_ = lambda l, c: mix_tau(l, c, maxl)  # noqa: E731
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
            s[f'cg_model.cormorant_cg.edge_levels.{k}.cat_mix.weights.{l}'] = (CH, edge_cin_(k, l, CH), 2)
    for k in range(NLEV):
        cout = co if k == NLEV - 1 else CH
        for l in range(MAXL + 1):
            s[f'cg_model.cormorant_cg.atom_levels.{k}.cat_mix.weights.{l}'] = (cout, atom_tau_(k, l, CH), 2)
    for l in range(MAXL + 1):
        s[f'cg_mix.cat_mix.weights.{l}'] = (CE, mix_tau_(l, CE), 2)
    for name, n_in, n_out in (('phi_focus', nlat, 1), ('phi_element', nlat, num_zs),
                              ('phi_d', nlat_e, 2 * num_gaussians), ('phi_trans', nlat, width), ('phi_v', width, 1)):
        s[f'{name}.layers.0.weight'] = (width, n_in)
        s[f'{name}.layers.0.bias'] = (width, )
        s[f'{name}.layers.1.weight'] = (n_out, width)
        s[f'{name}.layers.1.bias'] = (n_out, )
    s['distance_log_stds'] = (num_gaussians, )
    return s


def offsets(num_zs: int, width: int, num_gaussians: int, ch: int = CH, ce: int = CE, nlev: int = NLEV, maxl: int = MAXL):
    """name -> (offset, shape); also returns the total length."""
    out, off = OrderedDict(), 0
    for name, shape in slots(num_zs, width, num_gaussians, ch, ce, nlev, maxl).items():
        n = 1
        for d in shape:
            n *= d
        out[name] = (off, shape)
        off += n
    return out, off


def embedding_index(num_zs: int, width: int, num_gaussians: int, ch: int, ce: int, nlev: int, maxl: int):
    """For maxl < 4: position in the maxl = 4 parameter vector (the layout the kernels read) of every element of the
    maxl-limited one, as a list of ints.  The smaller network IS the larger one with every quantity of a degree above maxl
    absent: a missing input column or output row of a weight corresponds to a zero in the larger weight, so the embedded
    parameters (zeros elsewhere) give the same outputs, and the gradient of the smaller network is the gather of the larger one's
    at these positions.  Column orders (cormorant, as SURVEY Appendix A / oracle/encoder_ref.py have them):
      atom cat-mix, levels >= 1:  [aggregate blocks | input | power blocks], a block = ch channels, blocks in cg_blocks order;
      edge cat-mix, levels >= 1:  [previous edge net | DotMatrix, ch per degree l' <= maxl | radial];
      mixer:                      [aggregate | power blocks | input], ce channels each;
      first layers of the heads:  AtomicScalars features, (maxl + 2) blocks of 2 tau floats, block-major -> a prefix."""
    import numpy as np
    small, ns = offsets(num_zs, width, num_gaussians, ch, ce, nlev, maxl)
    big, _ = offsets(num_zs, width, num_gaussians, ch, ce, nlev, MAXL)
    idx = np.full(ns, -1, dtype=np.int64)

    def cols_blocks(l, first, tau, parts):
        """column map of one cat: parts = sequence of ('blocks' | 'one'): small column -> big column"""
        sb, bb = cg_blocks(l, maxl), cg_blocks(l, MAXL)
        m, so, bo = {}, 0, 0
        for kind in parts:
            if kind == 'one':
                for c in range(tau):
                    m[so + c] = bo + c
                so, bo = so + tau, bo + tau
            else:
                for j, blk in enumerate(sb):
                    jb = bb.index(blk)
                    for c in range(tau):
                        m[so + j * tau + c] = bo + jb * tau + c
                so, bo = so + len(sb) * tau, bo + len(bb) * tau
        return m

    for name, (off, shape) in small.items():
        boff, bshape = big[name]
        n = int(np.prod(shape))
        if tuple(shape) == tuple(bshape):
            idx[off:off + n] = boff + np.arange(n)
            continue
        rows, cols_s, cols_b = shape[0], shape[1], bshape[1]
        assert bshape[0] == rows and len(shape) == len(bshape)
        tail = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        if '.atom_levels.' in name:
            l = int(name.rsplit('.', 1)[1])
            cmap = cols_blocks(l, 0, ch, ('blocks', 'one', 'blocks'))
        elif '.edge_levels.' in name:
            cmap = {c: c for c in range((maxl + 2) * ch)}                           # previous | DotMatrix degrees <= maxl
            cmap.update({(maxl + 2) * ch + c: (MAXL + 2) * ch + c for c in range(ch)})  # radial
        elif name.startswith('cg_mix.'):
            l = int(name.rsplit('.', 1)[1])
            cmap = cols_blocks(l, 0, ce, ('one', 'blocks', 'one'))
        else:  # first layer of a head MLP on AtomicScalars features: a prefix of the larger feature vector
            assert name.endswith('layers.0.weight') and cols_s < cols_b, name
            cmap = {c: c for c in range(cols_s)}
        assert len(cmap) == cols_s and max(cmap.values()) < cols_b, name
        col_b = np.array([cmap[c] for c in range(cols_s)], dtype=np.int64)
        r = np.arange(rows)[:, None, None]
        t = np.arange(tail)[None, None, :]
        idx[off:off + n] = (boff + (r * cols_b + col_b[None, :, None]) * tail + t).reshape(-1)
    assert (idx >= 0).all() and len(np.unique(idx)) == ns
    return idx


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
